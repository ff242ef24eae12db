#pragma once
#include <memory>

#include "search/pool.h"

struct mi_search {
    std::unique_ptr<cra::search::SearchPool> pool;
};
