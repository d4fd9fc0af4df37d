"""Controller protocol of the SOFA loop (sofacontrol/closed_loop_controller.py:140-170).

`ClosedLoopController` itself is a `Sofa.Core.Controller` living inside the simulator and stays in the
reference; it only needs an object with this protocol: `evaluate(sim_time, y, x, u_prev) -> u`,
`set_sim_timestep(dt)`, `save_controller_info()`, attribute `observer.z`."""


class TemplateController:
    def __init__(self):
        pass

    def save_controller_info(self):
        return dict()

    def evaluate(self, time, y, x, u_prev):
        raise NotImplementedError('TemplateController must be subclassed')

    def set_sim_timestep(self, dt):
        self.sim_dt = dt
