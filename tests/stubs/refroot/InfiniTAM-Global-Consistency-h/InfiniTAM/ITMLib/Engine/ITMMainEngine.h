// where the reference expects the InfiniTAM submodule's header: forwards to the mirror (see ../../../../../README.md)
#pragma once
#include "ITMLib/Engine/ITMMainEngine.h"
