"""unsloth_amd: an MI355X-native (gfx950 / CDNA4) fused-kernel fine-tuning hot path behind unsloth's API.

    from unsloth_amd import FastLanguageModel
"""
__version__ = "0.1.0"


def __getattr__(name):          # lazy: importing the package must not pull transformers in
    if name in ("FastLanguageModel", "FastModel", "FastVisionModel", "is_bfloat16_supported"):
        from . import models
        return getattr(models, name)
    if name in ("UnslothTrainer", "UnslothTrainingArguments", "unsloth_train"):
        from . import trainer
        return getattr(trainer, name)
    raise AttributeError(name)
