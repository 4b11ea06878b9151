"""CPU oracle for the hot path (TEST INFRASTRUCTURE — never imported by the product).

A functional, dtype-generic restatement (plain torch ops on CPU, fp32 or fp64) of the
reference's encoder stack, MFN delta-memory gate and the three sequence models that
call them.  It exists so that the HIP path can be checked on the GPU box, where the
reference itself cannot travel.

Pinning: the reference ships no tests for this path, so the oracle
is pinned by fixtures captured in the build container from the imported reference
classes: ``tests/golden/make_golden.py`` (generator) -> ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` replays them on CPU.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.
"""
from .encoder_ref import (  # noqa: F401
    layer_norm, scaled_dot_attention, multi_head_attention, feed_forward,
    encoder_layer, encoder_stack, count_layers,
)
from .mfn_ref import lstm_cell, mfn_gate  # noqa: F401
from .models_ref import (  # noqa: F401
    multi_transformer, nlp_transformer, uni_full_transformer, lstm_decoder_head,
)
from .metrics import eval_ccc, masked_mse_sum_loss  # noqa: F401
from .frontend_ref import (  # noqa: F401
    cnn_maxpool, highway, window_encoder, multi_cnn_transformer_sft, multi_cnn_transformer_mft, multi_cnn_transformer_b2,
)
