/* cv stub (oracle/refbuild): everything lives in opencv2/core/core.hpp */
#include "../core/core.hpp"
