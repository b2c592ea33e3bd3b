// evaluate.h — mirror of kami/evaluate.h: the gating match between the current and the candidate model.
#pragma once

#include "nn/nn.h"

namespace kami {
    bool eval(NN* current_model, NN* candidate_model, int trainer);
}
