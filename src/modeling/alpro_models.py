"""`src.modeling.alpro_models` -> alpro_amd.modeling.alpro_models (same class names and signatures)."""
from alpro_amd.modeling.alpro_models import (AlproBaseModel, AlproForPretrain, AlproForSequenceClassification,  # noqa: F401
                                             AlproForVideoTextRetrieval, Prompter)
