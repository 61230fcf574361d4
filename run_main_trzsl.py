"""Entry point with the reference's name (run_main_trzsl.py:1-4 -> methods.main_TRZSL.main()) on the native engine."""
import grip_amd  # noqa: F401
from grip_amd.methods.main import main

if __name__ == "__main__":
    main("trzsl")
