// encode_square.h — Env::observe (kami/env.h:202-262) for ONE point-of-view square, shared by the
// stand-alone encode kernel (encode.hip) and the fused ingest of the tower kernel (tower_mfma.hip).
#pragma once
#include "../../include/kami_hip.h"

namespace kh {

// v[0..29] = the 30 channel values of POV square `povsq` (0..63) of record r.
__device__ __forceinline__ void encode_square(const kh_board* r, int povsq, float (&v)[KH_NFEATURES])
{
    const int ply = r->ply, hmc = r->halfmove_clock;
    const int ctm = r->ctm & 1, castle = r->castle_rights;
    // real square seen at POV square povsq (env.h:246: povsq = 63 - sq for black)
    const int sq = ctm ? 63 - povsq : povsq;
    // piece plane index 0..11 relative to channel 18, or -1 for an empty square
    int idx = -1;
#pragma unroll
    for (int t = 0; t < 6; ++t)
        if ((r->piece_occ[t] >> sq) & 1) idx = t;
    const int is_w = (int)((r->color_occ[0] >> sq) & 1);
    const int is_b = (int)((r->color_occ[1] >> sq) & 1);
    if (!(is_w | is_b)) idx = -1;
    // ncPieceColor(pc) != our_col -> +6 (env.h:255-256)
    if (idx >= 0 && (is_b != ctm)) idx += 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)((ply >> i) & 1);                 // env.h:213-214
#pragma unroll
    for (int i = 0; i < 6; ++i) v[8 + i] = (float)((hmc >> i) & 1);             // env.h:216-218
    // raw masked castle bits; black swaps the white/black pairs (env.h:220-236)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[14 + i] = (float)(castle & (1 << (i ^ (ctm << 1))));
#pragma unroll
    for (int i = 0; i < 12; ++i) v[18 + i] = (idx == i) ? 1.0f : 0.0f;         // env.h:258
}

}  // namespace kh
