from .base_model import BaseModel, WeightRepresentation, LevelRepresentation  # noqa: F401
from .mlp import MLPNet  # noqa: F401
from .vgg import VGG  # noqa: F401
from .vit import ViT  # noqa: F401
