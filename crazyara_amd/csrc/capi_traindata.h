#pragma once
#include "rl/traindata.h"

struct mi_traindata {
    cra::rl::TrainDataExporter exp;
    mi_traindata(const char* path, int mode, int vmaj, int vmin, unsigned chunks, unsigned chunk_size)
        : exp(path, mode, vmaj, vmin, chunks, chunk_size) {}
};
