"""grip_amd -- MI355X-native CLIP prompt-tuning + pseudolabel engine.

The directory is named `menghini-neurips23-code_amd/` (not a Python identifier);
import it as `grip_amd` through the `grip_amd.py` shim at the repository root.
Sub-packages mirror the reference's module names for the hot path only
(SURVEY.md section 8): `clip` (the surface of openai-CLIP the reference touches),
`models` (clip_encoders / prompts_models), `utils` (clip_pseudolabels),
`methods` (training-strategy stand-in).  The arithmetic lives in `csrc/`
(hand-written gfx950 HIP behind the C ABI declared in include/grip_amd.h).
"""
__version__ = "0.1.0"
