"""Drop-in for the surface of the third-party `clip` package that the reference uses
(`import clip`, `from clip import clip`): load, tokenize, model.Transformer, model.CLIP."""
from . import clip, model  # noqa: F401
from .clip import available_models, load, tokenize  # noqa: F401
