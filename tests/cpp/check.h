// Minimal test harness for the C++ host classes (no gtest in this image).
#pragma once
#include <cmath>
#include <cstdio>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

struct TestCase { const char* name; std::function<void()> fn; };
inline std::vector<TestCase>& registry() { static std::vector<TestCase> r; return r; }
struct Registrar { Registrar(const char* n, std::function<void()> f) { registry().push_back({n, f}); } };
#define TEST(name) static void name(); static Registrar reg_##name(#name, name); static void name()
#define CHECK(cond) do { if (!(cond)) throw std::runtime_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": CHECK(" #cond ") failed"); } while (0)
#define CHECK_THROWS(expr, Ex) do { bool ok_ = false; try { expr; } catch (const Ex&) { ok_ = true; } catch (...) {} if (!ok_) throw std::runtime_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": expected " #Ex); } while (0)
inline int run_all(const char* only = nullptr) {
    int failed = 0;
    std::setvbuf(stdout, nullptr, _IONBF, 0);
    for (auto& t : registry()) {
        if (only && std::string(t.name) != only) continue;
        std::printf("[ RUN] %s\n", t.name);
        try { t.fn(); std::printf("[ OK ] %s\n", t.name); }
        catch (const std::exception& e) { ++failed; std::printf("[FAIL] %s: %s\n", t.name, e.what()); }
    }
    std::printf("%zu tests, %d failed\n", registry().size(), failed);
    return failed ? 1 : 0;
}
