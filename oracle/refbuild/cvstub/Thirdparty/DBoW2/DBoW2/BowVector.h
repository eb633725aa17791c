/* stand-in: BowVector lives in FeatureVector.h of this stub (oracle/_ref test infrastructure) */
#include "FeatureVector.h"
