#!/bin/bash
# Developer tool: build the DIAGNOSTIC library tools/bin/libmmt_abl.so — the product sources plus -DMMT_ABLATIONS: the stamped twins of the
# attention forward and of the one-kernel attention backward, generated from the product headers (tools/make_diag.py).
set -e
cd "$(dirname "$0")/.."
python3 tools/make_diag.py multimodal_transformer_amd/csrc/attn_bwd_pair.h tools/bin/gen/attn_bwd_pair_diag.h attn_bwd_pair16_kernel > /dev/null
python3 tools/make_diag.py multimodal_transformer_amd/csrc/attn.h tools/bin/gen/attn_fwd_diag.h attn_fwd_kernel > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-unused-value -DMMT_ABLATIONS \
    -Itools/bin/gen -o tools/bin/libmmt_abl.so multimodal_transformer_amd/csrc/api.hip
echo built tools/bin/libmmt_abl.so
