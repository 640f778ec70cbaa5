// corr_d4.hip -- tuned correlation kernels (placeholder until the first tuned version lands)
#include "common.h"
namespace cerb {
int corr_d4_forward(const void *, const void *, void *, const CorrGeom &, float, int64_t, int,
                    hipStream_t) { return CERB_EUNSUPPORTED; }
int corr_d4_backward(const void *, const void *, const void *, void *, void *, const CorrGeom &,
                     int, hipStream_t) { return CERB_EUNSUPPORTED; }
}  // namespace cerb
