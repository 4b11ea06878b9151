#!/bin/bash
# Developer tool: build an EXPERIMENT variant of the library from a patched copy of csrc/ (the product sources stay untouched).
#   tools/build_exp.sh <name> <sed-script-file or python patch script>   -> tools/bin/libmmt_<name>.so
# The patch script is run as `python3 <script> <dir>` and edits the copied sources in place.
set -e
cd "$(dirname "$0")/.."
NAME=$1; PATCH=$2; shift 2
D=/tmp/mmt_exp_$NAME
rm -rf $D && mkdir -p $D && cp -r multimodal_transformer_amd/csrc $D/csrc && mkdir -p $D/include && cp include/mmt_hip.h $D/include/
sed -i 's#"../../include/mmt_hip.h"#"../include/mmt_hip.h"#' $D/csrc/api.hip
python3 $PATCH $D/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-unused-value "$@" -o tools/bin/libmmt_$NAME.so $D/csrc/api.hip
echo built tools/bin/libmmt_$NAME.so
