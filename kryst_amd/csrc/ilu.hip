// ILU(0)-family preconditioners: placeholder until the level-scheduled triangular solve lands (next commit).
#include "pc.h"
namespace kr {
int32_t ilu_apply_dev(kryst_pc_t, const double*, double*, const int*) { set_error("ILU apply not built yet"); return KRYST_UNSUPPORTED; }
void ilu_free(kryst_pc_t) {}
}
extern "C" int32_t kryst_pc_ilu0(kryst_csr_t, int32_t, kryst_pc_t*) { kr::set_error("ILU setup not built yet"); return KRYST_UNSUPPORTED; }
