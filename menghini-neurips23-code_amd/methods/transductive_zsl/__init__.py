from ..training_strategies import TrainingStrategy  # noqa: F401
from ..strategies import MultimodalFPL, MultimodalPrompt, TextualFPL, TextualPrompt, VisualFPL, VisualPrompt  # noqa: F401
