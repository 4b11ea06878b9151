#!/usr/bin/env python3
"""Developer tool: turn the `//@phase` marker comments of a COPY of csrc/rowgemm.h into per-workgroup cycle stamps (tools/build_phase.sh
patches a copy of csrc/ with this script and builds tools/bin/libmmt_phase.so from it; the product header holds no diagnostic code).
Stamps are accumulated in registers of thread 0 and flushed once per stage into the workgroup's PRIVATE slots with plain stores: an atomic
per mark costs ~7k cycles, and contended atomics at the end of a stage make the next stage wait ~25k cycles for them — both artefacts were
larger than the phases they were meant to measure.  Read out by tools/phase_timing.py."""
import re
import sys

p = sys.argv[1] + "/rowgemm.h"
s = open(p).read()
MACROS = r'''
__device__ unsigned long long* g_phase_buf = nullptr;      // [workgroup][stage slot 0..3][phase 0..7] accumulated cycles
#define PHASE_DECL unsigned long long t_phase_ = __builtin_readcyclecounter(), acc_phase_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
    const int slot_phase_ = (EPI == EPI_FRAG) ? 0 : (EPI == EPI_LNBWD ? 1 : (LNPRO ? 2 : 3));
#define PHASE(n) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); \
    acc_phase_[n] += now_ - t_phase_; t_phase_ = now_; } } while (0)
#define PHASE_FLUSH do { if (threadIdx.x == 0 && g_phase_buf) { unsigned long long* q_ = g_phase_buf + ((size_t)blockIdx.x * 4 + slot_phase_) * 8; \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) if (acc_phase_[i_]) q_[i_] += acc_phase_[i_]; } } while (0)
'''
assert '#include "common.h"' in s
s = s.replace('#include "common.h"', '#include "common.h"' + MACROS, 1)
s = s.replace("//@phase decl", "PHASE_DECL")
s = s.replace("//@phase flush", "PHASE_FLUSH;")
s = re.sub(r"/\*@phase (\d)\*/", r"PHASE(\1);", s)
s = re.sub(r"//@phase (\d):", r"PHASE(\1);   //", s)
assert "@phase" not in s.replace("`//@phase`", "")
open(p, "w").write(s)
