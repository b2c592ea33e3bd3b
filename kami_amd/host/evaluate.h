// evaluate.h — the gating match between the serving model and a freshly trained candidate
// (kami/evaluate.h; implementation in evaluate.cpp over this repository's search).
#ifndef KAMI_AMD_HOST_EVALUATE_H
#define KAMI_AMD_HOST_EVALUATE_H

#include "nn/nn.h"

namespace kami {
// true: the candidate scored at least "evaluate_target_pct" percent and replaces the current model
bool eval(NN* current_model, NN* candidate_model, int trainer);
}  // namespace kami
#endif
