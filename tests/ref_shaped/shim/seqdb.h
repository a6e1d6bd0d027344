// see myutils.h in this directory (include-path shim, test infrastructure)
#pragma once
#include "myutils.h"
