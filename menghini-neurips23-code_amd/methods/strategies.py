"""The reference's strategy class names (methods/<paradigm>/{textual,visual,multimodal}_{prompt,fpl}.py)
as thin aliases of the generic TrainingStrategy: the class fixes the prompt modality and whether FPL
pseudolabels are merged into the training set; the paradigm comes from config.LEARNING_PARADIGM."""
from .training_strategies import TrainingStrategy


def _make(name, modality, fpl, takes_data_folder):
    def __init__(self, config, label_to_idx, *args):
        # reference signatures: (config, label_to_idx, [data_folder, [unlabeled_files,]] classes, seen, unseen, device)
        *rest, classes, seen, unseen, device = args
        config.MODALITY = modality
        TrainingStrategy.__init__(self, config, label_to_idx, classes, seen, unseen, device, data_folder=rest[0] if rest else None)
    return type(name, (TrainingStrategy,), {"__init__": __init__, "fpl": fpl, "__doc__": f"{name}: modality={modality}, fpl={fpl}"})


TextualPrompt = _make("TextualPrompt", "text", False, False)
VisualPrompt = _make("VisualPrompt", "image", False, False)
MultimodalPrompt = _make("MultimodalPrompt", "multi", False, False)
TextualFPL = _make("TextualFPL", "text", True, True)
VisualFPL = _make("VisualFPL", "image", True, True)
MultimodalFPL = _make("MultimodalFPL", "multi", True, True)
