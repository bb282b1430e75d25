// mr_tuning: defaults, validation, the MEGREADER_TUNING environment variable (see include/megreader_hip.h).
#include <climits>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"
#include "tuning.h"

namespace mr {

#define MR_TUNING_DEFAULTS                                                                                              \
  { /*nt_variant*/ 2, /*nt_deep*/ 1, /*nt_big*/ 0, /*nt_p8*/ 0, /*nt_force_bm*/ 0, /*nt_force_bn*/ 0, /*gemm_skinny*/ 1, \
    /*tn_big*/ 0, /*tn_buf*/ 1, /*tn_taps*/ 1, /*tn_taps_group*/ 0, /*tn_group*/ 0, /*tn_fin*/ 0, /*tn_taps_fin*/ 0,     \
    /*tn_taps_w8*/ 0, /*tn_model*/ 1, /*tn_splits*/ 0, /*bn_fused*/ 1, /*lstm_persist*/ 1, /*lstm_fwd_bn*/ 0,            \
    /*lstm_bwd_bn*/ 32, /*dcn_fused*/ 1, /*dcn_v1_bwd*/ 1, /*bn_onepass*/ 1, /*skinny_depth*/ 0, /*nt_big_min_k*/ 512, /*tn_taps_min_p*/ 10000, /*tn_defer*/ 1, /*pool_fixed*/ 1, /*ctc_linear*/ 1, /*nt_wide8*/ 7, /*nt_ksplit*/ 1, /*nt_m32*/ 0, /*nt_m32_opt*/ 0, /*dcn_gcol*/ 1, /*dcn_col_fwd*/ 1, /*decode_persist*/ 1, {0} }

mr_tuning g_tuning = MR_TUNING_DEFAULTS;
static const mr_tuning k_defaults = MR_TUNING_DEFAULTS;
static std::mutex g_tuning_mutex;   // writers only; readers load single fields atomically (MR_TUNE)

struct Field {
  const char* name;
  size_t offset;
  int lo, hi;   // accepted range
};
#define MR_F(f, lo, hi) {#f, offsetof(mr_tuning, f), lo, hi}
static const Field k_fields[] = {
    MR_F(nt_variant, 1, 2), MR_F(nt_deep, 0, 2), MR_F(nt_big, -1, 13),
#ifdef MR_ABLATION
    MR_F(nt_p8, 0, 4),   // 2..4: timing-only ablation variants (wrong results), tools build only
#else
    MR_F(nt_p8, 0, 1),
#endif
    MR_F(nt_force_bm, 0, 128), MR_F(nt_force_bn, 0, 128), MR_F(gemm_skinny, 0, 1), MR_F(tn_big, -1, 2), MR_F(tn_buf, 0, 1),
    MR_F(tn_taps, 0, 1), MR_F(tn_taps_group, 0, 1 << 20), MR_F(tn_group, 0, 1 << 20), MR_F(tn_fin, 0, 2),
    MR_F(tn_taps_fin, 0, 2), MR_F(tn_taps_w8, 0, 1), MR_F(tn_model, 0, 1), MR_F(tn_splits, 0, 1 << 20), MR_F(bn_fused, 0, 1),
    MR_F(lstm_persist, 0, 2), MR_F(lstm_fwd_bn, 0, 64), MR_F(lstm_bwd_bn, 0, 64), MR_F(dcn_fused, 0, 1),
    MR_F(dcn_v1_bwd, 0, 1), MR_F(bn_onepass, 0, 1), MR_F(skinny_depth, 0, 8), MR_F(nt_big_min_k, 32, 1 << 20), MR_F(tn_taps_min_p, 0, 1 << 30), MR_F(tn_defer, 0, 1), MR_F(pool_fixed, 0, 1), MR_F(ctc_linear, 0, 1), MR_F(nt_wide8, 0, 63), MR_F(nt_ksplit, 0, 8), MR_F(nt_m32, 0, 5), MR_F(nt_m32_opt, 0, 99), MR_F(dcn_gcol, 0, 1), MR_F(dcn_col_fwd, 0, 1), MR_F(decode_persist, 0, 2)};
#undef MR_F

static const char* check(const mr_tuning& t) {
  for (const Field& f : k_fields) {
    const int v = *(const int*)((const char*)&t + f.offset);
    if (v < f.lo || v > f.hi) return f.name;
  }
  if (t.tn_fin == 1) return "tn_fin";
  if (t.nt_force_bm != 0 && !((t.nt_force_bm == 128 || t.nt_force_bm == 96 || t.nt_force_bm == 64) &&
                              (t.nt_force_bn == 128 || t.nt_force_bn == 64)))
    return "nt_force_bm / nt_force_bn";
  if (!(t.skinny_depth == 0 || t.skinny_depth == 4 || t.skinny_depth == 8)) return "skinny_depth";
  if (!(t.lstm_fwd_bn == 0 || t.lstm_fwd_bn == 32 || t.lstm_fwd_bn == 64)) return "lstm_fwd_bn";
  if (!(t.lstm_bwd_bn == 0 || t.lstm_bwd_bn == 16 || t.lstm_bwd_bn == 32 || t.lstm_bwd_bn == 64)) return "lstm_bwd_bn";
  return nullptr;
}

static void store(const mr_tuning& t) {   // field by field: a concurrent reader sees old or new values, never a torn int
  for (const Field& f : k_fields)
    __atomic_store_n((int*)((char*)&g_tuning + f.offset), *(const int*)((const char*)&t + f.offset), __ATOMIC_RELAXED);
}

// MEGREADER_TUNING="name=value,name=value": applied once by mr_init (before any launch of a normal program)
int tuning_from_env() {
  static bool done = false;
  std::lock_guard<std::mutex> lock(g_tuning_mutex);
  if (done) return MR_OK;
  done = true;
  const char* env = std::getenv("MEGREADER_TUNING");
  if (!env || !*env) return MR_OK;
  mr_tuning t = g_tuning;
  char buf[1024];
  std::strncpy(buf, env, sizeof(buf) - 1);
  buf[sizeof(buf) - 1] = 0;
  for (char* tok = std::strtok(buf, ",; "); tok; tok = std::strtok(nullptr, ",; ")) {
    char* eq = std::strchr(tok, '=');
    if (!eq) { set_error("MEGREADER_TUNING: '%s' is not name=value", tok); return MR_ERR_ARG; }
    *eq = 0;
    bool found = false;
    for (const Field& f : k_fields)
      if (!std::strcmp(f.name, tok)) {
        *(int*)((char*)&t + f.offset) = std::atoi(eq + 1);
        found = true;
      }
    if (!found) { set_error("MEGREADER_TUNING: unknown field '%s'", tok); return MR_ERR_ARG; }
  }
  if (const char* bad = check(t)) { set_error("MEGREADER_TUNING: value of '%s' out of range", bad); return MR_ERR_ARG; }
  store(t);
  return MR_OK;
}

}  // namespace mr

using namespace mr;

extern "C" {

int mr_tuning_get(mr_tuning* out) {
  MR_CHECK_ARG(out != nullptr, "mr_tuning_get: null");
  std::lock_guard<std::mutex> lock(g_tuning_mutex);
  *out = g_tuning;
  return MR_OK;
}

int mr_tuning_defaults(mr_tuning* out) {
  MR_CHECK_ARG(out != nullptr, "mr_tuning_defaults: null");
  *out = k_defaults;
  return MR_OK;
}

int mr_tuning_set(const mr_tuning* in) {
  MR_CHECK_ARG(in != nullptr, "mr_tuning_set: null");
  if (const char* bad = check(*in)) {
    set_error("mr_tuning_set: value of '%s' out of range", bad);
    return MR_ERR_ARG;
  }
  std::lock_guard<std::mutex> lock(g_tuning_mutex);
  store(*in);
  return MR_OK;
}

}  // extern "C"
