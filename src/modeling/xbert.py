"""`src.modeling.xbert` -> alpro_amd.modeling.xbert (the classes ALPRO instantiates)."""
from alpro_amd.modeling.xbert import (BertAttention, BertEmbeddings, BertEncoder, BertForMaskedLM, BertIntermediate,  # noqa: F401
                                      BertLayer, BertLMPredictionHead, BertModel, BertOnlyMLMHead, BertOutput,
                                      BertPredictionHeadTransform, BertPreTrainedModel, BertSelfAttention, BertSelfOutput)
