from .loader import FastLanguageModel, FastModel, FastVisionModel
from .llama import FastLlamaModel


def is_bfloat16_supported():
    """unsloth/models/__init__.py:30 re-export; MI355X always has bf16 MFMA."""
    return True
