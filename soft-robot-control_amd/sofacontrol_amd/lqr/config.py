"""iLQR settings (values of sofacontrol/lqr/config.py:1-31)."""


class iLQRConfig:
    def __init__(self):
        self.max_iter = 50
        self.epsilon = 0.1
        self.include_input_var_constraint = True
        self.do_linesearch = True
        self.regularize = True
        # line search (forward pass)
        self.alpha0 = 1.
        self.alpha_scaling = 0.5
        self.improv_lb = 1e-4
        self.improv_ub = 100
        self.alpha_min = 5e-2
        self.counter_limit = 5
        # regularisation (backward pass)
        self.rho0 = 0.
        self.drho0 = 0.
        self.rho_scaling = 1.5
        self.rho_increase_fp = 10.
        self.rho_max = 1e5
        self.rho_min = 1e-3
        self.state_regularization = True
