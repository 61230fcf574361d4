from .strategies import MultimodalFPL, MultimodalPrompt, TextualFPL, TextualPrompt, VisualFPL, VisualPrompt  # noqa: F401
from .clip_baseline import ClipBaseline  # noqa: F401
from .training_strategies import TrainingStrategy  # noqa: F401
