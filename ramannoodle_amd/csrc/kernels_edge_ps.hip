// placeholder (filled in below in this round)
#include "fused_common.hpp"
namespace rn {}
