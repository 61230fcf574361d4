"""ORACLE (test infrastructure) -- CPU stand-in for the third-party `clip`
package the reference imports (`import clip`, `from clip import clip`)."""
from . import clip, model  # noqa: F401
from .clip import available_models, load, tokenize  # noqa: F401
