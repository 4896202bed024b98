"""MI355X-native forward/backward for the ASY-VRNet (Efficient-VRNet) fusion hot path."""
from .init_utils import randomize_state_dict, synthetic_inputs  # noqa: F401
from .net import EfficientVRNet  # noqa: F401
