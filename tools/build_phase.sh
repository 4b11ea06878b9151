#!/bin/bash
# Developer tool: tools/bin/libmmt_phase.so — the library built from a patched copy of csrc/ in which rowgemm.h's `//@phase` markers are cycle
# stamps (tools/make_phase.py), with -DMMT_PHASE_TIMING (api.hip then exports mmt_debug_set_phase_buffer).  Read out by tools/phase_timing.py.
exec "$(dirname "$0")/build_exp.sh" phase "$(dirname "$0")/make_phase.py" -DMMT_PHASE_TIMING
