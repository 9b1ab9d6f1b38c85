"""Moved to tests/golden/det_init.py (round 3): the closed-form parameter / batch generator is fixture infrastructure, not part of the
oracle -- bench.py's in-run parity check regenerates the fixtures' weights and inputs from it without touching anything under oracle/.
This module keeps the old import path working for the tests and tools written against it."""
from tests.golden.det_init import *  # noqa: F401,F403
from tests.golden.det_init import (PromptEncoding, _splitmix64, canonical_name, det_batch, det_caption_ids, det_param, det_prompts, det_raw_clips, fill_state_dict_,  # noqa: F401
                                   unit_uniform)
