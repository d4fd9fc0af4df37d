"""Target containers (sofacontrol/tpwl/tpwl_utils.py:5-38)."""
from .. import utils as scutils


class Target:
    def __init__(self):
        self.t = None
        self.u = None
        self.z = None
        self.x = None
        self.Hf = None

    def load_target_file(self, file):
        data = scutils.load_data(file)
        self.t = data.get('t')
        self.u = data.get('u')
        self.z = data.get('z')
        self.Hf = data.get('Hf')


class DynamicsTarget(Target):
    def __init__(self):
        super().__init__()
        self.A = None
        self.B = None
        self.x = None
