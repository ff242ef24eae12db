#pragma once
#include "nn/rise_net.h"

struct mi_net {
    cra::RiseNet net;
    mi_net(const char* dir, int dev, int batch, const char* prec) : net(dir ? dir : "", dev, batch, prec ? prec : "float16") {}
};
