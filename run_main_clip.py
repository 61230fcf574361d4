"""Entry point with the reference's name (run_main_clip.py -> methods.main_CLIP.main()): zero-shot CLIP on the native engine."""
import grip_amd  # noqa: F401
from grip_amd.methods.main import main

if __name__ == "__main__":
    main("ul", model="clip_baseline")
