// Entry points orc_post_error / orc_post_backward / orc_sse_* of oracle/currennt_oracle.c routed to the six translation
// units built from ref_post.cpp (one per reference layer file).  No reference code in this file.
#include "ref_common.h"
typedef float real_t;
#define DECL(k) \
    extern "C" real_t ref_post_error_##k(int, int, const char *, const real_t *, const real_t *); \
    extern "C" void ref_post_backward_##k(int, int, const char *, const real_t *, const real_t *, real_t *);
DECL(sse) DECL(weightedsse) DECL(wf) DECL(ce) DECL(rmse) DECL(binary)

REF_API real_t ref_post_error(int kind, int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs)
{
    switch (kind) {
    case 0: return ref_post_error_sse(L, N, patTypes, targets, outputs);
    case 1: return ref_post_error_weightedsse(L, N, patTypes, targets, outputs);
    case 2: return ref_post_error_wf(L, N, patTypes, targets, outputs);
    case 3: return ref_post_error_ce(L, N, patTypes, targets, outputs);
    case 4: return ref_post_error_rmse(L, N, patTypes, targets, outputs);
    default: return ref_post_error_binary(L, N, patTypes, targets, outputs);
    }
}

REF_API void ref_post_backward(int kind, int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs, real_t *outErr)
{
    switch (kind) {
    case 0: ref_post_backward_sse(L, N, patTypes, targets, outputs, outErr); break;
    case 1: ref_post_backward_weightedsse(L, N, patTypes, targets, outputs, outErr); break;
    case 2: ref_post_backward_wf(L, N, patTypes, targets, outputs, outErr); break;
    case 3: ref_post_backward_ce(L, N, patTypes, targets, outputs, outErr); break;
    case 4: ref_post_backward_rmse(L, N, patTypes, targets, outputs, outErr); break;
    default: ref_post_backward_binary(L, N, patTypes, targets, outputs, outErr); break;
    }
}

REF_API real_t ref_sse_error(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs)
{
    return ref_post_error_sse(L, N, patTypes, targets, outputs);
}

REF_API void ref_sse_backward(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs, real_t *outErr)
{
    ref_post_backward_sse(L, N, patTypes, targets, outputs, outErr);
}
