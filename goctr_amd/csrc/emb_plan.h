// emb_plan.h -- interface of emb_plan.hip (the plan build of the trainable-embedding extension; its own translation unit
// because it pulls in rocPRIM's radix sort).
#pragma once
#include <cstdint>

namespace goctr {

struct EmbPlanSource {            // the id-mode dataset the plan is built for (device pointers)
  const int32_t* ub_ids; const int32_t* item_ids; long long rows; long long V;
};
struct EmbPlanArrays {            // device arrays sized by the caller: pairs <= nb B (T + 1), slots <= nb min(B (T + 1), V)
  int* pair; int* pslot; int* pid;           // [pairs]
  int* slot_id; unsigned int* slot_off;      // [slots], [slots + nb]
  long long* pair_off; long long* slot_base; // [nb + 1]
};
// builds the plan of all nb batches on the calling thread's engine; totals_host = {pairs, slots, max pairs per batch, max slots
// per batch}.  Synchronises the engine stream once, at the end.
int emb_plan_build(const EmbPlanSource& src, int B, int T, int W, long long Vw, long long nb, const EmbPlanArrays& out, long long totals_host[4]);

}  // namespace goctr
