// w2v.hip -- item2vec engine (float64).  TEMPORARY stubs: every entry point fails loudly.
#include "common.h"
using namespace goctr;
#define NOTYET(name) do { set_error(name ": not implemented in this build"); return -1; } while (0)
extern "C" {
void goctr_w2v_cfg_default(goctr_w2v_cfg* c) { memset(c, 0, sizeof *c); }
int goctr_w2v_create(const goctr_w2v_cfg*, int64_t, const int64_t*, goctr_w2v**) { NOTYET("goctr_w2v_create"); }
void goctr_w2v_destroy(goctr_w2v*) {}
int goctr_w2v_set_param(goctr_w2v*, const double*) { NOTYET("goctr_w2v_set_param"); }
int goctr_w2v_set_aux(goctr_w2v*, const double*) { NOTYET("goctr_w2v_set_aux"); }
int goctr_w2v_get_param(goctr_w2v*, double*) { NOTYET("goctr_w2v_get_param"); }
int goctr_w2v_get_aux(goctr_w2v*, double*) { NOTYET("goctr_w2v_get_aux"); }
int goctr_w2v_get_paths(goctr_w2v*, int64_t*, int32_t*, uint8_t*, int64_t, int64_t*) { NOTYET("goctr_w2v_get_paths"); }
int goctr_w2v_train(goctr_w2v*, const int32_t*, int64_t, int64_t, const uint8_t*, double*) { NOTYET("goctr_w2v_train"); }
int goctr_w2v_upload_doc(goctr_w2v*, const int32_t*, int64_t, const uint8_t*) { NOTYET("goctr_w2v_upload_doc"); }
int goctr_w2v_train_resident(goctr_w2v*, int64_t, double*) { NOTYET("goctr_w2v_train_resident"); }
int goctr_w2v_export_f32(goctr_w2v*, float*) { NOTYET("goctr_w2v_export_f32"); }
}
