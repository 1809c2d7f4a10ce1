"""track_mjx.environment mirror: batched rodent tracking env over the HIP C-ABI."""
from .task import MultiClipTracking, RewardConfig, State, METRIC_NAMES  # noqa: F401
from .wrappers import wrap  # noqa: F401
from .reward import compute_tracking_rewards  # noqa: F401
