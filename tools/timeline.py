"""Kernel timeline of the last complete step in a rocprofv3 --kernel-trace csv (all streams):
    python tools/timeline.py <kernel_trace.csv> [first-kernel-prefix]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'chomp::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].split('(')[0].replace('chomp::', '').replace('void ', '')
first = sys.argv[2] if len(sys.argv) > 2 else 'k_proj_chi'
starts = [i for i, r in enumerate(rows) if name(r).startswith(first)]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]['Start_Timestamp'])
end = t0
print('%-36s %9s %9s %9s  %s' % ('kernel', 'start us', 'dur us', 'end us', 'queue / grid x wg'))
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-36s %9.1f %9.1f %9.1f  q%s %s x %s' % (name(r), (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3,
                                                 r.get('Queue_Id', '?'), r['Grid_Size_X'], r['Workgroup_Size_X']))
    end = max(end, e)
print('step: first start -> last end %.1f us; -> next step start %.1f us' % (
    (end - t0) / 1e3, (int(rows[b]['Start_Timestamp']) - t0) / 1e3))
