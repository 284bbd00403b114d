// kmg_internal.h -- what the translation units of libkmeans_hip share besides the kernels' launchers: the error
// channel of the C ABI and the few host helpers of kmg_api.hip / kmg_processor.hip that the multi-device layer (kmg_group.hip) reuses.
// Nothing here is exported (-fvisibility=hidden).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <array>
#include <vector>

#include "../../include/kmeans_hip.h"

// Tuning switches and knock-outs (kernel variants that skip work and return WRONG results, for measurements) exist only in the
// tools build (make tools: -DKMG_TOOLS, lib/libkmeans_hip_tools.so, loaded by tools/ through KMG_LIBRARY).  In the product
// library the environment is not even consulted for them: KMG_TOOLS_ENV is a macro so that the variable's name is not in the
// binary, and KMG_KNOCK(flags, bit) is a constant so that the knocked-out paths are not compiled.
#ifdef KMG_TOOLS
#include <stdlib.h>
#define KMG_TOOLS_ENV(name) getenv(name)
#define KMG_KNOCK(flags, bit) (((flags) & (bit)) != 0u)
#else
#define KMG_TOOLS_ENV(name) (static_cast<const char *>(nullptr))
#define KMG_KNOCK(flags, bit) (false)
#endif

#include <stdlib.h>
static inline int tools_env_int(const char *e, int dflt) { return e ? atoi(e) : dflt; }

namespace kmg {

// sets the calling thread's kmg_last_error() message and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
bool log_debug();

// No C++ exception leaves the library: the callers of this ABI are C, Rust (rust-shim/) and ctypes, for which an exception that
// unwinds through an extern "C" frame is undefined behaviour, where the reference returns anyhow::Result (core/src/lib.rs:38).
// Every extern "C" definition is a function-try-block closed by one of the macros below; abi_trap() names what was caught:
// std::bad_alloc -> KMG_ERR_OUT_OF_MEMORY, any other std::exception -> KMG_ERR_HIP with what() in kmg_last_error(), anything
// else -> KMG_ERR_HIP.  (tests/test_abi.py checks that no definition is left out.)
int abi_trap() noexcept;
#define KMG_ABI_CATCH      catch (...) { return kmg::abi_trap(); }
#define KMG_ABI_CATCH_VOID catch (...) { (void)kmg::abi_trap(); }
#define KMG_ABI_CATCH_NULL catch (...) { (void)kmg::abi_trap(); return nullptr; }

// lib.rs:255-286 kmeans_palette: CentroidsBuffer::pull_values (palette-crate Lab -> sRGB8) of a k-means centroid table, then
// sorted ascending by the palette-crate Lab L of the 8-bit colour.  out_rgba: k x 4 bytes.
void sorted_palette_of(const float *centroids4, uint32_t k, uint8_t *out_rgba);
// lib.rs:288-331 octree_palette on a host image already shrunk to <= 128: the reference's CPU octree, sorted by L
std::vector<std::array<uint8_t, 4>> octree_sorted_palette(const uint8_t *host_rgba, uint64_t n_pixels, uint32_t color_count);

// An image between a caller's (pageable) buffer and the device, ordered on `st` (kmg_api.hip): small images asynchronously,
// large ones as synchronous row-range copies on several streams of the processor.
hipError_t copy_host_image(kmg_processor *p, void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st);

}  // namespace kmg
