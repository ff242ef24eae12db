#pragma once
#include <exception>
#include <string>

void cra_set_error(const std::string& msg);
const char* cra_get_error();

template <typename F> int cra_guard(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        cra_set_error(e.what());
    } catch (...) {
        cra_set_error("unknown error");
    }
    return 1;
}
