// The library's environment switches, parsed once (mf_switches.hpp says what each one is for).
#include "mf_switches.hpp"
#include <cstdlib>

namespace mf {

static bool env_set(const char *name) { return getenv(name) != nullptr; }
static bool env_is(const char *name, char first) { const char *e = getenv(name); return e && e[0] == first; }
static long long env_ll(const char *name, long long dflt) { const char *e = getenv(name); return e ? strtoll(e, nullptr, 0) : dflt; }

Switches switches_parse() {
    Switches s;
    // diagnostics: always honoured (they print, they do not change what runs)
    s.verbose = env_set("MF_VERBOSE");
    s.chain_verbose = env_set("MF_CHAIN_VERBOSE");
    s.dq_verbose = env_set("MF_DQ_VERBOSE");
    s.debug_epi = env_set("MF_DEBUG_EPI");
    // everything below changes routing or tuning: the product ignores it unless the process says MF_DEV=1
    s.dev = env_is("MF_DEV", '1');
    if (!s.dev) return s;
    s.no_rt = env_set("MF_NO_RT");
    s.no_stem_rt = env_set("MF_NO_STEM_RT");
    s.no_chain = env_set("MF_NO_CHAIN");
    s.chain_no_sp = env_set("MF_CHAIN_NO_SP");
    s.no_stage = env_set("MF_NO_STAGE");
    s.no_dwfc = env_set("MF_NO_DWFC");
    s.no_pairtail = env_set("MF_NO_PAIRTAIL");
    s.no_quad = env_set("MF_NO_QUAD");
    s.no_pair_front = env_set("MF_NO_PAIR_FRONT");
    s.no_quad_mm = env_set("MF_NO_QUAD_MM");
    s.no_penta = env_set("MF_NO_PENTA");
    s.no_f32_group = env_set("MF_NO_F32_GROUP");
    s.quads = (int)env_ll("MF_QUADS", 7);
    s.no_magic = env_set("MF_NO_MAGIC");
    s.no_sat_pack = env_set("MF_NO_SAT_PACK");
    s.no_table = env_set("MF_NO_TABLE");
    s.chain_all = env_set("MF_CHAIN_ALL");
    s.chain_force = env_set("MF_CHAIN_FORCE");
    s.chain_no_res = env_set("MF_CHAIN_NO_RES");
    s.conv_mm_256 = env_set("MF_CONV_MM_256");
    s.no_fma_epi = env_set("MF_NO_FMA_EPI");
    s.no_fast_quant_div = env_set("MF_NO_FAST_QUANT_DIV");
    s.dwpw_mm_only = env_is("MF_DWPW_IMPL", 'm');
    s.stem_valu = env_is("MF_STEM_IMPL", 'v');
    s.dw_c1_lds = env_is("MF_DW_C1", 'l');
    s.dw_alt = (int)env_ll("MF_DW_ALT", -1);
    s.dwmm_alt = (int)env_ll("MF_DWMM_ALT", -1);
    s.dwrr_alt = (int)env_ll("MF_DWRR_ALT", -1);
    s.fc_tile = (int)env_ll("MF_FC_TILE", 0);
    s.fc_rowsum_fold = env_set("MF_FC_ROWSUM_FOLD");
    s.fc_rowsum_prepass = env_set("MF_FC_ROWSUM_PREPASS");
    s.pw_grid = env_ll("MF_PW_GRID", 0);
    s.pw_rt_ncap = (int)env_ll("MF_PW_RT_NCAP", 64);
    s.dw_rt_threads = (int)env_ll("MF_DW_RT_THREADS", 0);
    s.chain_autotune = !env_set("MF_CHAIN_AUTOTUNE") ? -1 : (env_is("MF_CHAIN_AUTOTUNE", '0') ? 0 : 1);
    s.chain_tune_g = !env_is("MF_CHAIN_TUNE_G", '0');
    if (const char *e = getenv("MF_CHAIN_OPCOST")) s.chain_opcost = atof(e);
    s.chain_dq_auto = env_set("MF_CHAIN_DQ_AUTO");
    s.dq_tune = env_set("MF_DQ_TUNE");
    s.dq_cfg_set = env_set("MF_DQ_CFG");
    s.dq_cfg = (int)env_ll("MF_DQ_CFG", 0);
    if (const char *l = getenv("MF_DQ_CFGS")) s.dq_cfgs = l;
    if (const char *e = getenv("MF_DQ_STATIC")) s.dq_static = atof(e);
    return s;
}

const Switches &switches() {
    static const Switches s = switches_parse();
    return s;
}

} // namespace mf
