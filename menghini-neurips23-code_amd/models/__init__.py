from .clip_encoders import CustomImageEncoder, CustomTextEncoder, ImageEncoder, TextEncoder  # noqa: F401
from .prompts_models import ImagePrefixModel, TextPrefixModel, UPTModel  # noqa: F401
