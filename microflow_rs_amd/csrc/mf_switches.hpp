// Every switch libmicroflow_amd.so reads from the environment, in one place.
//
// None of them is part of the product interface (that is include/microflow_amd.h), and the library IGNORES every routing / tuning
// switch unless the process also sets MF_DEV=1 (a stray MF_NO_QUAD in somebody's shell must not change which kernels a product
// runs); only the diagnostics (MF_VERBOSE, MF_CHAIN_VERBOSE, MF_DQ_VERBOSE, MF_DEBUG_EPI) work without it.  They are the A/B levers of the test matrix
// (scripts/switch_matrix.sh: the parity suite must be green under every MF_NO_* switch, because each turns one fused kernel off
// and sends its operators down the next more general path) and of the tuning scripts (scripts/tune_*.sh, scripts/dq_*.py).
// The environment is read ONCE, at the first call of switches(); the only exception is the step-queue tuning set, re-read per
// launch while MF_DQ_TUNE is set (scripts/dq_sweep.py changes it between launches of one process).
#pragma once
#include <string>

namespace mf {

struct Switches {
    bool dev = false;              // MF_DEV=1            master switch: without it every field below the diagnostics keeps its default
    // ---- routing: turn one specialised path off (tests: results must not change) ----
    bool no_rt = false;            // MF_NO_RT            no run-time-shape kernels (k_rt.hip): generated shapes take the generic kernels
    bool no_stem_rt = false;       // MF_NO_STEM_RT       ... for the C = 1 stem only
    bool no_chain = false;         // MF_NO_CHAIN         no run-time-shape pair chains (k_chain.hip)
    bool chain_no_sp = false;      // MF_CHAIN_NO_SP      no spatially split stride-2 head in a chain
    bool no_stage = false;         // MF_NO_STAGE         no 6x6x128 four-pair stage kernel (k_stage.hip)
    bool no_dwfc = false;          // MF_NO_DWFC          no depthwise + FC + softmax kernel (k_dwfc.hip, speech)
    bool no_pairtail = false;      // MF_NO_PAIRTAIL      no pair + pool + conv + softmax tail (k_tail3.hip)
    bool no_pair_front = false;    // MF_NO_PAIR_FRONT    ops 23..24 as their own dwpw_mm launch, not inside the pair + tail launch (k_tail3.hip FRONT)
    bool no_quad_mm = false;       // MF_NO_QUAD_MM       ops 9..12 as two dwpw_mm launches, not one quad_mm launch (k_quad_mm.hip)
    bool no_quad = false;          // MF_NO_QUAD          no two-pair register-resident kernels (k_quad.hip)
    bool no_penta = false;         // MF_NO_PENTA         no stem + two pairs kernel (k_quad.hip)
    bool no_f32_group = false;     // MF_NO_F32_GROUP     the f32 entry quantises in the stem kernel, not inside the stem + two pairs launch
    int quads = 7;                 // MF_QUADS            bit mask of the quad shapes allowed (1 | 2 | 4)
    bool no_magic = false;         // MF_NO_MAGIC         requantisation by v_cvt (epilogue mode 0) everywhere
    bool no_sat_pack = false;      // MF_NO_SAT_PACK      clamp by v_med3 (mode 1), never v_sat_pk (mode 2)
    bool no_table = false;         // MF_NO_TABLE         table shapes take the run-time-geometry kernels too
    bool chain_all = false;        // MF_CHAIN_ALL        the chain kernel on table shapes too
    bool chain_force = false;      // MF_CHAIN_FORCE      the planner never prefers the unfused operators
    bool chain_no_res = false;     // MF_CHAIN_NO_RES     chains reload their operands every step (no register residency)
    bool conv_mm_256 = false;      // MF_CONV_MM_256      round 3's four-wave workgroups for the run-time conv kernel
    bool no_fma_epi = false;       // MF_NO_FMA_EPI       keep the two-rounding requantisation (epilogue modes 1/2) everywhere
    bool no_fast_quant_div = false;// MF_NO_FAST_QUANT_DIV  boundary quantisation by IEEE division, not the verified 3-instruction form
    bool dwpw_mm_only = false;     // MF_DWPW_IMPL=mm     every pair on dwpw_mm (LDS intermediate), none on dwpw_rr (registers)
    bool stem_valu = false;        // MF_STEM_IMPL=valu   stem taps on the VALU kernel (dw3x3_stem8), not the matrix-pipe one
    bool dw_c1_lds = false;        // MF_DW_C1=lds        C = 1 depthwise through the LDS-tiled kernel
    // ---- tuning candidates (scripts/tune_*.sh; -1 / 0 = the table's choice) ----
    int dw_alt = -1;               // MF_DW_ALT           candidate index of the layer-wise depthwise launch table
    int dwmm_alt = -1;             // MF_DWMM_ALT         ... of dwpw_mm
    int dwrr_alt = -1;             // MF_DWRR_ALT         ... of dwpw_rr
    int fc_tile = 0;               // MF_FC_TILE          force the FullyConnected tile (1..)
    bool fc_rowsum_fold = false;   // MF_FC_ROWSUM_FOLD   the row sums of a weight zero point from v_dot4 between the GEMM's MFMAs (round 4's form)
    bool fc_rowsum_prepass = false;// MF_FC_ROWSUM_PREPASS  ... from the fc_rowsum launch in front of the GEMM (round 1's form), not the GEMM's own prologue
    long long pw_grid = 0;         // MF_PW_GRID          force the pointwise grid
    int pw_rt_ncap = 64;           // MF_PW_RT_NCAP       widest N the run-time pointwise kernel takes
    int dw_rt_threads = 0;         // MF_DW_RT_THREADS    force the run-time depthwise workgroup size
    // ---- chain planner ----
    int chain_autotune = -1;       // MF_CHAIN_AUTOTUNE   -1 unset (the handle's flag decides), 0 off, 1 on for every handle
    bool chain_tune_g = true;      // MF_CHAIN_TUNE_G=0   autotune keeps the model's images-per-step
    double chain_opcost = 1.0;     // MF_CHAIN_OPCOST     weight of the operand-reload term of the chain cost model
    bool chain_dq_auto = false;    // MF_CHAIN_DQ_AUTO    chains take the automatic step-queue configuration even for short launches
    // ---- step queue (k_common.hpp dq_config) ----
    bool dq_tune = false;          // MF_DQ_TUNE          re-read MF_DQ_CFG / MF_DQ_CFGS at every launch
    bool dq_cfg_set = false;       // MF_DQ_CFG present
    int dq_cfg = 0;                // MF_DQ_CFG           K | heads << 8 | static << 16 for every launch
    std::string dq_cfgs;           // MF_DQ_CFGS          "c0,c1,...": configuration of the k-th queue launch of the process
    double dq_static = 0.6;        // MF_DQ_STATIC        share of a workgroup's steps walked by stride before the queue deals
    // ---- diagnostics ----
    bool verbose = false;          // MF_VERBOSE
    bool chain_verbose = false;    // MF_CHAIN_VERBOSE
    bool dq_verbose = false;       // MF_DQ_VERBOSE
    bool debug_epi = false;        // MF_DEBUG_EPI        which epilogue mode each operator gets, and why
};

Switches switches_parse();          // reads the environment now
const Switches &switches();         // parsed at first use, then fixed for the life of the process

} // namespace mf
