"""Device-side input pipeline: the reference's host augmentation chain as HIP kernels + a small host planner."""
from .augment import AugmentHyp, DeviceAugmenter, plan_train_sample, train_labels  # noqa: F401
from .loader import DeviceAugmentLoader  # noqa: F401
