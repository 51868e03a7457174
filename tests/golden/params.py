"""Parameter dictionaries shared by the golden generator and the tests.
Values are the test inputs of the reference's unit_test.py:59-119 (data)."""
c_dict = {"omega_m0": 0.3 - 4.15e-5 / 0.7 ** 2, "omega_b0": 0.046, "omega_l0": 0.7,
          "omega_r0": 4.15e-5 / 0.7 ** 2, "cmb_temp": 2.726, "h": 0.7,
          "sigma_8": 0.8, "n_scalar": 0.960, "w0": -1.0, "wa": 0.0}
c_dict_2 = dict(c_dict, omega_m0=1.0 - 4.15e-5 / 0.7 ** 2, omega_l0=0.0)
h_dict_2 = {"stq": 0.5, "st_little_a": 0.5, "c0": 5., "beta": -0.2, "alpha": -1,
            "delta_v": 200.0}
hod_dict = {"log_M_min": 12.14, "sigma": 0.15, "log_M_0": 12.14,
            "log_M_1p": 13.43, "alpha": 1.0}
hod_dict_2 = {"log_M_min": 14.06, "sigma": 0.71, "log_M_0": 14.06,
              "log_M_1p": 14.80, "alpha": 1.0}
