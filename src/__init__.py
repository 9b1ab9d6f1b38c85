"""Import-path shim: the reference's drivers do `from src.modeling.alpro_models import ...`
(run_pretrain_sparse.py:29, run_video_retrieval.py:26).  These modules re-export the MI355X
implementation under the same dotted paths so those imports resolve unchanged."""
