/* include-path switch for oracle/refbuild (TEST INFRASTRUCTURE): `#include "ORBextractor.h"` resolves to the PRODUCT's
 * extractor shim while every other header name keeps resolving to the reference's include/ directory. */
#include "../../../orb_slam2_ssd_semantic_amd/shim/ORBextractor.h"
