// fccqp_fused_f32.hip -- the shape-specialised kernel for Q / A_eq stored as float (the
// fp32-storage entry points of include/fccqp.h): same source as fccqp_fused.hip, own translation
// unit (see the note there).
#define FCCQP_FUSED_STORAGE_F32 1
#include "fccqp_fused.hip"
