// placeholder: the fused step kernel is under construction; plan calls fail loudly.
#include "fused.h"
namespace wv {
struct FusedPlan { int dummy; };
FusedPlan *fused_create(const Grid &, const float *, const float *) { return new FusedPlan{0}; }
void fused_destroy(FusedPlan *p) { delete p; }
void fused_set_pml(FusedPlan *, const float *, const float *) {}
void fused_state_changed(FusedPlan *) {}
void fused_state_zeroed(FusedPlan *) {}
int fused_energy_blocks(const FusedPlan *) { return 1; }
int fused_prepare(FusedPlan *, const float *, const Cyl *, const Cyl *, int, int, hipStream_t) { return 1; }
void fused_launch(FusedPlan *, const FusedStep &, hipStream_t) {}
}  // namespace wv
