// Host-side thread-safety check of libmvn_hip's process-global state (tests/test_tsan_host.py::test_host_threads_under_tsan).
// Built together with csrc/mvn_hip.hip, host code only, under the host ThreadSanitizer; needs no GPU: eight threads call entry
// points that validate their arguments, query the dispatcher (the MVN_* switch table, the CU-count cache) and ask for the
// dynamic-LDS opt-in table, while one of them keeps re-reading the switches.  ThreadSanitizer reports any unsynchronised
// access (exit code 66); the test also checks that the answers are the single-threaded ones.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/mvn.h"

int main() {
    char expect_decode[128], expect_sweep[128], expect_train[128];
    if (mvn_vnet_decode_kernel_name(10000, 1000, 16, 0, expect_decode, sizeof expect_decode)) return 2;
    if (mvn_acs_sweep_kernel_name(nullptr, nullptr, 1000, 10000, 1000, 16, expect_sweep, sizeof expect_sweep)) return 2;
    if (mvn_vnet_train_kernel_name(2, 200, 136, 1, 16, (size_t)1 << 30, expect_train, sizeof expect_train)) return 2;
    std::atomic<int> bad{0};
    std::vector<std::thread> threads;
    for (int i = 0; i < 8; ++i)
        threads.emplace_back([&, i] {
            char name[128];
            for (int k = 0; k < 3000; ++k) {
                if (mvn_acs_sweep_f32(nullptr, nullptr, 8, nullptr, 4, 8, 3, nullptr) != MVN_E_STATES) ++bad;
                if (mvn_va_decode_f32(nullptr, 8, nullptr, 3, nullptr, 8, nullptr, 4, 8, 16, nullptr) != MVN_E_PRIORS) ++bad;
                if (mvn_vnet_decode_kernel_name(10000, 1000, 16, 0, name, sizeof name) || strcmp(name, expect_decode)) ++bad;
                if (mvn_acs_sweep_kernel_name(nullptr, nullptr, 1000, 10000, 1000, 16, name, sizeof name) || strcmp(name, expect_sweep)) ++bad;
                if (mvn_vnet_train_kernel_name(2, 200, 136, 1, 16, (size_t)1 << 30, name, sizeof name) || strcmp(name, expect_train)) ++bad;
                if (mvn_vnet_workspace_bytes(10, 100, 16) != 0) ++bad;
                // a launch path up to its first device call: the step kernel asks for its dynamic-LDS opt-in (fails without a
                // device, after walking the opt-in table under its mutex)
                float dummy[8];
                (void)mvn_vnet_byword_step_f32(dummy, 136, dummy, 120, dummy, dummy, dummy, dummy, dummy, dummy, nullptr, nullptr, 136,
                                               nullptr, 120, nullptr, 136, nullptr, 136, nullptr, 136, nullptr, 1, 136, 2, 1, 16, nullptr);
                if (i == 0 && k % 50 == 0) mvn_reload_switches();
            }
        });
    for (auto &t : threads) t.join();
    std::printf("tsan driver: %d wrong answers\n", bad.load());
    return bad.load() ? 1 : 0;
}
