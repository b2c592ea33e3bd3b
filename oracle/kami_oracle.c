/*
 * kami_oracle.c — CPU restatement of kami's leaf-evaluation path.  TEST INFRASTRUCTURE
 * (see kami_oracle.h for who may use it and how it is pinned).
 *
 * Plain C99 + OpenMP over boards.  Follows the reference line by line in behaviour,
 * not in code: the reference's arithmetic lives in libtorch (unpinned; the build
 * container has torch 2.10.0 CPU), whose published semantics are restated here:
 *   Conv2d   = cross-correlation, zero padding 1 for 3x3, bias added          (nn.cpp:20-21,45-51)
 *   BatchNorm2d (eval) = (x - running_mean) / sqrt(running_var + 1e-5) * weight + bias
 *   Linear   = x W^T + b                                                      (nn.cpp:52)
 *   exp(log_softmax(x)) with max subtraction                                  (nn.cpp:80)
 */
#include "kami_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PSIZE 4672
#define PPLANES 73
#define PMID 128
#define VW 256
#define NFEAT 30

/* ------------------------------------------------------------------ encoding */

/* kami/env.h:202-262.  Channel map:
 *   [0..7]   bits of history.size(), LSB first            env.h:211-214
 *   [8..13]  bits of the half-move clock                  env.h:216-218
 *   [14..17] castle_rights & {our K, our Q, opp K, opp Q} RAW masked value (1,2,4,8) env.h:220-236
 *   [18..23] our P,N,B,R,Q,K   [24..29] opponent's        env.h:241-260
 *   black to move => square index mirrored, povsq = 63 - sq   env.h:246            */
void ko_observe(const ko_board* b, float* dst)
{
    float header[18];
    int our = b->ctm & 1;
    memset(dst, 0, sizeof(float) * 64 * NFEAT);
    for (int i = 0; i < 8; ++i) header[i] = (float)((b->ply >> i) & 1);
    for (int i = 0; i < 6; ++i) header[8 + i] = (float)((b->halfmove_clock >> i) & 1);
    int our_k = 1, our_q = 2, opp_k = 4, opp_q = 8;          /* position.h:13-16 */
    if (our == 1) { our_k = 4; our_q = 8; opp_k = 1; opp_q = 2; }
    header[14] = (float)(b->castle_rights & our_k);
    header[15] = (float)(b->castle_rights & our_q);
    header[16] = (float)(b->castle_rights & opp_k);
    header[17] = (float)(b->castle_rights & opp_q);
    for (int sq = 0; sq < 64; ++sq) memcpy(dst + sq * NFEAT, header, sizeof header);
    for (int sq = 0; sq < 64; ++sq) {
        uint64_t m = 1ULL << sq;
        int col;
        if (b->color_occ[0] & m) col = 0;
        else if (b->color_occ[1] & m) col = 1;
        else continue;
        int t = -1;
        for (int k = 0; k < 6; ++k) if (b->piece_occ[k] & m) { t = k; break; }
        if (t < 0) continue;
        int povsq = our ? 63 - sq : sq;
        float* base = dst + NFEAT * povsq + 18;
        if (col != our) base += 6;
        base[t] = 1.0f;
    }
}

void ko_observe_batch(const ko_board* b, int n, float* dst)
{
    for (int i = 0; i < n; ++i) ko_observe(b + i, dst + (size_t)i * 64 * NFEAT);
}

/* FEN as printed by ncPositionToFen (position.c:131-165) / ncBoardToFen (board.c:142-176):
 * ranks 8..1, files a..h, pieces "PpNnBbRrQqKk", then ctm, castling "KQkq"/"-", ep, hmc, fullmove. */
int ko_board_from_fen(const char* fen, int ply, ko_board* out)
{
    memset(out, 0, sizeof *out);
    int r = 7, f = 0;
    const char* p = fen;
    for (; *p && *p != ' '; ++p) {
        char c = *p;
        if (c == '/') { --r; f = 0; continue; }
        if (c >= '1' && c <= '8') { f += c - '0'; continue; }
        const char* pcs = "PpNnBbRrQqKk";
        const char* q = strchr(pcs, c);
        if (!q || r < 0 || f > 7) return 1;
        int pc = (int)(q - pcs);
        int sq = r * 8 + f;
        out->piece_occ[pc >> 1] |= 1ULL << sq;
        out->color_occ[pc & 1] |= 1ULL << sq;
        ++f;
    }
    if (*p != ' ') return 2;
    ++p;
    out->ctm = (*p == 'b') ? 1 : 0;
    p += 2;
    for (; *p && *p != ' '; ++p) {
        if (*p == 'K') out->castle_rights |= 1;
        if (*p == 'Q') out->castle_rights |= 2;
        if (*p == 'k') out->castle_rights |= 4;
        if (*p == 'q') out->castle_rights |= 8;
    }
    if (*p != ' ') return 3;
    ++p;
    while (*p && *p != ' ') ++p;          /* en passant: not observed */
    if (*p != ' ') return 4;
    out->halfmove_clock = atoi(p + 1);
    out->ply = ply;
    return 0;
}

/* ------------------------------------------------------------------ weights */

typedef struct { const float *w, *b, *g, *be, *rm, *rv; } convbn;

typedef struct {
    int F, C, R;
    convbn stem;
    convbn* res;            /* 2*R */
    convbn pconv;           /* C -> 128 */
    const float *p2w, *p2b; /* 128 -> 73 */
    convbn vconv;           /* C -> 1 */
    const float *fcw, *fcb; /* 64 -> 256 */
} net;

size_t ko_weight_count(int F, int C, int R)
{
    size_t n = 0;
    n += (size_t)C * F * 9 + C + 4 * (size_t)C;
    n += (size_t)R * 2 * ((size_t)C * C * 9 + C + 4 * (size_t)C);
    n += (size_t)PMID * C + PMID + 4 * PMID;
    n += (size_t)PPLANES * PMID + PPLANES;
    n += (size_t)C + 1 + 4;
    n += (size_t)VW * 64 + VW;
    return n;
}

static const float* take(const float** p, size_t n) { const float* r = *p; *p += n; return r; }

static void take_convbn(const float** p, convbn* c, size_t wn, int co)
{
    c->w = take(p, wn); c->b = take(p, co);
    c->g = take(p, co); c->be = take(p, co); c->rm = take(p, co); c->rv = take(p, co);
}

static int net_parse(net* n, const float* blob, int F, int C, int R)
{
    const float* p = blob;
    n->F = F; n->C = C; n->R = R;
    take_convbn(&p, &n->stem, (size_t)C * F * 9, C);
    n->res = (convbn*)malloc(sizeof(convbn) * 2 * (size_t)(R > 0 ? R : 1));
    if (!n->res) return 1;
    for (int i = 0; i < 2 * R; ++i) take_convbn(&p, &n->res[i], (size_t)C * C * 9, C);
    take_convbn(&p, &n->pconv, (size_t)PMID * C, PMID);
    n->p2w = take(&p, (size_t)PPLANES * PMID); n->p2b = take(&p, PPLANES);
    take_convbn(&p, &n->vconv, (size_t)C, 1);
    n->fcw = take(&p, (size_t)VW * 64); n->fcb = take(&p, VW);
    return 0;
}

/* ------------------------------------------------------------------ layers */

/* 3x3 conv (pad 1, bias) + eval BN + ReLU on one board, channels-last activations.
 * wt is the weight re-laid as [tap][ci][co] so the inner loop runs over co. */
static void conv3_bn_relu(const float* x, int Ci, int Co, const float* wt, const convbn* c,
                          float* y, float* acc)
{
    for (int py = 0; py < 8; ++py)
        for (int px = 0; px < 8; ++px) {
            for (int co = 0; co < Co; ++co) acc[co] = 0.0f;
            for (int ky = 0; ky < 3; ++ky) {
                int iy = py + ky - 1;
                if (iy < 0 || iy > 7) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    int ix = px + kx - 1;
                    if (ix < 0 || ix > 7) continue;
                    const float* xi = x + (size_t)(iy * 8 + ix) * Ci;
                    const float* wk = wt + (size_t)(ky * 3 + kx) * Ci * Co;
                    for (int ci = 0; ci < Ci; ++ci) {
                        float a = xi[ci];
                        const float* wr = wk + (size_t)ci * Co;
                        for (int co = 0; co < Co; ++co) acc[co] += a * wr[co];
                    }
                }
            }
            float* yo = y + (size_t)(py * 8 + px) * Co;
            for (int co = 0; co < Co; ++co) {
                float v = acc[co] + c->b[co];
                v = (v - c->rm[co]) / sqrtf(c->rv[co] + 1e-5f) * c->g[co] + c->be[co];
                yo[co] = v < 0.0f ? 0.0f : v;   /* NaN propagates, like torch::relu */
            }
        }
}

static float* relayout3(const float* w, int Co, int Ci)
{
    /* libtorch [Co][Ci][3][3] -> [tap][Ci][Co] */
    float* t = (float*)malloc(sizeof(float) * 9 * (size_t)Ci * Co);
    if (!t) return NULL;
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int k = 0; k < 9; ++k)
                t[((size_t)k * Ci + ci) * Co + co] = w[((size_t)co * Ci + ci) * 9 + k];
    return t;
}

int ko_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int ko_forward(const float* blob, int F, int C, int R, const float* in, int B,
               float* policy, float* value_full, float* logits, int nthreads)
{
    net n;
    if (net_parse(&n, blob, F, C, R)) return 1;
    int rc = 0;
    float* wstem = relayout3(n.stem.w, C, F);
    float** wres = (float**)calloc((size_t)(2 * R > 0 ? 2 * R : 1), sizeof(float*));
    for (int i = 0; i < 2 * R; ++i) wres[i] = relayout3(n.res[i].w, C, C);
    /* policyconv [128][C] -> [C][128]; policyconv2 [73][128] -> [128][73] */
    float* wp1 = (float*)malloc(sizeof(float) * (size_t)C * PMID);
    float* wp2 = (float*)malloc(sizeof(float) * PMID * PPLANES);
    for (int co = 0; co < PMID; ++co) for (int ci = 0; ci < C; ++ci) wp1[(size_t)ci * PMID + co] = n.pconv.w[(size_t)co * C + ci];
    for (int co = 0; co < PPLANES; ++co) for (int ci = 0; ci < PMID; ++ci) wp2[(size_t)ci * PPLANES + co] = n.p2w[(size_t)co * PMID + ci];
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        int cm = C > PMID ? C : PMID;
        float* x = (float*)malloc(sizeof(float) * 64 * (size_t)C);
        float* t = (float*)malloc(sizeof(float) * 64 * (size_t)C);
        float* u = (float*)malloc(sizeof(float) * 64 * (size_t)C);
        float* acc = (float*)malloc(sizeof(float) * (size_t)cm);
        float* ph = (float*)malloc(sizeof(float) * 64 * PMID);
        float* lg = (float*)malloc(sizeof(float) * PSIZE);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            const float* xin = in + (size_t)b * 64 * F;
            /* stem: conv1 -> batchnorm1 -> relu            nn.cpp:62-65 */
            conv3_bn_relu(xin, F, C, wstem, &n.stem, x, acc);
            /* residual tower: x = x + relu(bn2(conv2(relu(bn1(conv1 x)))))   nn.cpp:26-34,68-69 */
            for (int r = 0; r < R; ++r) {
                conv3_bn_relu(x, C, C, wres[2 * r], &n.res[2 * r], t, acc);
                conv3_bn_relu(t, C, C, wres[2 * r + 1], &n.res[2 * r + 1], u, acc);
                for (int i = 0; i < 64 * C; ++i) x[i] = x[i] + u[i];
            }
            /* policy head: policyconv -> pbatchnorm -> relu -> policyconv2   nn.cpp:72-75 */
            for (int px = 0; px < 64; ++px) {
                for (int co = 0; co < PMID; ++co) acc[co] = 0.0f;
                for (int ci = 0; ci < C; ++ci) {
                    float a = x[(size_t)px * C + ci];
                    const float* wr = wp1 + (size_t)ci * PMID;
                    for (int co = 0; co < PMID; ++co) acc[co] += a * wr[co];
                }
                for (int co = 0; co < PMID; ++co) {
                    float v = acc[co] + n.pconv.b[co];
                    v = (v - n.pconv.rm[co]) / sqrtf(n.pconv.rv[co] + 1e-5f) * n.pconv.g[co] + n.pconv.be[co];
                    ph[px * PMID + co] = v < 0.0f ? 0.0f : v;
                }
            }
            /* permute({0,2,3,1}).flatten(1): index = pixel*73 + plane        nn.cpp:78-79 */
            for (int px = 0; px < 64; ++px) {
                float l[PPLANES];
                for (int co = 0; co < PPLANES; ++co) l[co] = 0.0f;
                for (int ci = 0; ci < PMID; ++ci) {
                    float a = ph[px * PMID + ci];
                    const float* wr = wp2 + ci * PPLANES;
                    for (int co = 0; co < PPLANES; ++co) l[co] += a * wr[co];
                }
                for (int co = 0; co < PPLANES; ++co) lg[px * PPLANES + co] = l[co] + n.p2b[co];
            }
            if (logits) memcpy(logits + (size_t)b * PSIZE, lg, sizeof(float) * PSIZE);
            /* exp(log_softmax(ph, 1))                                         nn.cpp:80 */
            {
                float m = lg[0];
                for (int i = 1; i < PSIZE; ++i) if (lg[i] > m || isnan(lg[i])) m = lg[i];
                double s = 0.0;
                for (int i = 0; i < PSIZE; ++i) s += exp((double)(lg[i] - m));
                float ls = (float)log(s);
                float* po = policy + (size_t)b * PSIZE;
                for (int i = 0; i < PSIZE; ++i) po[i] = expf((lg[i] - m) - ls);
            }
            /* value head: valueconv -> vbatchnorm -> relu -> flatten -> valuefc -> tanh   nn.cpp:83-88 */
            {
                float v64[64];
                for (int px = 0; px < 64; ++px) {
                    float a = 0.0f;
                    for (int ci = 0; ci < C; ++ci) a += x[(size_t)px * C + ci] * n.vconv.w[ci];
                    a += n.vconv.b[0];
                    a = (a - n.vconv.rm[0]) / sqrtf(n.vconv.rv[0] + 1e-5f) * n.vconv.g[0] + n.vconv.be[0];
                    v64[px] = a < 0.0f ? 0.0f : a;
                }
                float* vo = value_full + (size_t)b * VW;
                for (int j = 0; j < VW; ++j) {
                    float a = 0.0f;
                    const float* wr = n.fcw + (size_t)j * 64;
                    for (int k = 0; k < 64; ++k) a += v64[k] * wr[k];
                    vo[j] = tanhf(a + n.fcb[j]);
                }
            }
        }
        free(x); free(t); free(u); free(acc); free(ph); free(lg);
    }
    free(wstem);
    for (int i = 0; i < 2 * R; ++i) free(wres[i]);
    free(wres); free(wp1); free(wp2); free(n.res);
    return rc;
}

int ko_infer(const float* blob, int F, int C, int R, const float* in, int B,
             float* policy, float* value, int nthreads)
{
    float* vf = (float*)malloc(sizeof(float) * (size_t)B * VW);
    if (!vf) return 1;
    int rc = ko_forward(blob, F, C, R, in, B, policy, vf, NULL, nthreads);
    if (!rc) {
        /* nn.cpp:176-180: policy checked first, then value */
        for (size_t i = 0; i < (size_t)B * PSIZE; ++i) if (isnan(policy[i])) { rc = 4; break; }
        if (!rc) for (size_t i = 0; i < (size_t)B * VW; ++i) if (isnan(vf[i])) { rc = 5; break; }
        /* nn.cpp:186: memcpy(value, value_data, batch * sizeof(float)) on the [B,256] tensor */
        if (!rc) memcpy(value, vf, sizeof(float) * (size_t)B);
    }
    free(vf);
    return rc;
}
