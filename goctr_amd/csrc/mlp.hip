// mlp.hip -- sklearn-port MLP engine (float64).  TEMPORARY stubs: every entry point fails loudly.
#include "common.h"
using namespace goctr;
#define NOTYET(name) do { set_error(name ": not implemented in this build"); return -1; } while (0)
extern "C" {
void goctr_mlp_cfg_default(goctr_mlp_cfg* c) { memset(c, 0, sizeof *c); }
int goctr_mlp_create(const goctr_mlp_cfg*, goctr_mlp**) { NOTYET("goctr_mlp_create"); }
void goctr_mlp_destroy(goctr_mlp*) {}
size_t goctr_mlp_nparams(const goctr_mlp*) { return 0; }
int goctr_mlp_set_params(goctr_mlp*, const double*, size_t) { NOTYET("goctr_mlp_set_params"); }
int goctr_mlp_get_params(goctr_mlp*, double*, size_t) { NOTYET("goctr_mlp_get_params"); }
int goctr_mlp_loss_grad(goctr_mlp*, const double*, const double*, int, double*, double*) { NOTYET("goctr_mlp_loss_grad"); }
int goctr_mlp_fit(goctr_mlp*, const float*, const float*, int64_t, const int32_t*, double*, int*) { NOTYET("goctr_mlp_fit"); }
int goctr_mlp_upload(goctr_mlp*, const float*, const float*, int64_t) { NOTYET("goctr_mlp_upload"); }
int goctr_mlp_train_steps(goctr_mlp*, int64_t, int) { NOTYET("goctr_mlp_train_steps"); }
int goctr_mlp_predict(goctr_mlp*, const float*, int64_t, float*) { NOTYET("goctr_mlp_predict"); }
}
