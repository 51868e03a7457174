"""Default parameter dictionaries: same names, keys and values as the reference's
defaults.py (defaults.py:6-92), and like there they are module-level and read at
call time, so scripts that monkey-patch them (unit_test.py:17,
examples/shear_shear_spectrum.py:65-78) keep working.  A device context snapshots
default_limits / default_precision when it is created."""

default_cosmo_dict = {
    "omega_m0": 0.278 - 4.15e-5 / 0.7 ** 2,
    "omega_b0": 0.046,
    "omega_l0": 0.722,
    "omega_r0": 4.15e-5 / 0.7 ** 2,
    "cmb_temp": 2.726,
    "h": 0.7,
    "sigma_8": 0.811,
    "n_scalar": 0.960,
    "w0": -1.0,
    "wa": 0.0,
}

default_halo_dict = {
    "stq": 0.3,
    "st_little_a": 0.707,
    "c0": 9.0,
    "beta": -0.13,
    "alpha": -1,
    "delta_v": -1.0,
}

default_hod_dict = {
    "log_M_min": 12.14,
    "sigma": 0.15,
    "log_M_0": 12.14,
    "log_M_1p": 13.43,
    "alpha": 1.0,
}

default_limits = {
    "k_min": 0.001,
    "k_max": 100.0,
    "mass_min": -1,
    "mass_max": -1,
}

default_precision = {
    "corr_npoints": 50,
    "corr_precision": 1.48e-6,
    "cosmo_npoints": 50,
    "cosmo_precision": 1.48e-8,
    "dNdz_precision": 1.48e-8,
    "halo_npoints": 50,
    "halo_precision": 1.48e-5,
    "halo_limit": 100,
    "kernel_npoints": 50,
    "kernel_precision": 1.48e-6,
    "kernel_limit": 100,
    "kernel_bessel_limit": 8,
    "mass_npoints": 50,
    "mass_precision": 1.48e-8,
    "window_npoints": 100,
    "window_precision": 1.48e-6,
    "global_precision": 1.48e-32,
    "divmax": 20,
}
