// hap_forward.hip -- HaplotypeModel forward (model_dev.LSTMNetwork.predict,
// HaplotypeModel/model_dev.py:133-143).  Kernels land in the next commit; until then the entry
// points report NSNP_ENOWEIGHTS so that no caller can mistake a stub for a result.
#include "nsnp_common.hpp"

void nsnp_hap_free(nsnp_ctx*) {}

extern "C" int nsnp_hap_load_weights(nsnp_ctx* ctx, const float* const*, int, int, int, int, int, int)
{
    return ctx ? NSNP_ESHAPE : NSNP_EINVAL;
}

extern "C" int nsnp_hap_forward(nsnp_ctx* ctx, const float*, const float*, int64_t, float*, float*, void*)
{
    return ctx ? NSNP_ENOWEIGHTS : NSNP_EINVAL;
}
