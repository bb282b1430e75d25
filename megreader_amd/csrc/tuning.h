// Process-wide tuning / A-B state of the library: ONE struct (include/megreader_hip.h: mr_tuning), read through MR_TUNE(field)
// with a relaxed atomic load per field, replaced as a whole by mr_tuning_set().  The per-kernel `mr_set_*` functions of rounds
// 1-3 (26 unsynchronised globals behind 26 exported setters) are gone: the host wrappers below read the same switches from here.
#pragma once
#include "../../include/megreader_hip.h"

namespace mr {
extern mr_tuning g_tuning;   // tuning.hip
int tuning_from_env();       // MEGREADER_TUNING, applied once (called by mr_init)
}
#define MR_TUNE(f) (__atomic_load_n(&mr::g_tuning.f, __ATOMIC_RELAXED))
