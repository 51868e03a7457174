"""Per-kernel register / scratch / LDS table of a -save-temps build of the library:
    cd /tmp/x && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -save-temps -o lib.so <repo>/chomp_amd/csrc/chomp_capi.hip
    python tools/kernel_regs.py /tmp/x/chomp_capi-hip-amdgcn-amd-amdhsa-gfx950.s [filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in re.findall(r'(- \.agpr_count:.*?\.wavefront_size:\s+\d+)', s, re.S):
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk)
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    if flt and flt not in dn:
        continue
    print('%-58s vgpr %3s agpr %3s scratch %5s B  static lds %6s  sgpr %3s  vgpr spills %s' % (
        dn[-58:], g('vgpr_count').group(1), g('agpr_count').group(1),
        g('private_segment_fixed_size').group(1), g('group_segment_fixed_size').group(1),
        g('sgpr_count').group(1), g('vgpr_spill_count').group(1) if g('vgpr_spill_count') else '?'))
