from .git_rebasin import GitRebasin  # noqa: F401
from .otfusion import OTFusion  # noqa: F401
from .hf_kf import LSFusion  # noqa: F401
from .alf import ALFusion, ALFusionConfig, ActivationDataset  # noqa: F401
