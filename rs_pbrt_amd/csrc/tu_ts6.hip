// one group of kernel instantiations of librspt.so (tu_decl.h says which and why)
#include <hip/hip_runtime.h>
#include "../../include/rspt.h"
#define RSPT_TU_TEMPLATES_ONLY
#define RSPT_TU_X
#define RSPT_TU_GROUP_TS6
#include "tu_decl.h"
