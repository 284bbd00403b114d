// The multi-device layer of the C ABI (include/kmeans_hip.h, kmg_group_*) without Python or torch: the same image through ONE
// processor and through a group, results compared byte for byte.
//   check_group nogpu                       kmg_group_create must fail loudly without a device
//   check_group run <ranks> <k> <w> <h>     ranks == 1: one device, RCCL loaded and every collective issued
//                                           (KMG_GROUP_FORCE_COLLECTIVES); ranks > 1: that many ranks on device 0 through the
//                                           loopback exchange (RCCL refuses two ranks on one device)
// Exit code 0 and a last line "ok ..." on success.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kmeans_hip.h"

#define CHECK(expr)                                                                                          \
    do {                                                                                                     \
        const int rc_ = (expr);                                                                              \
        if (rc_ != KMG_OK) { fprintf(stderr, "%s -> %d: %s (line %d)\n", #expr, rc_, kmg_last_error(), __LINE__); return 1; } \
    } while (0)
#define HIPCHECK(expr)                                                                                       \
    do {                                                                                                     \
        const hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s (line %d)\n", #expr, hipGetErrorString(e_), __LINE__); return 1; } \
    } while (0)
#define EXPECT(cond)                                                                                         \
    do {                                                                                                     \
        if (!(cond)) { fprintf(stderr, "EXPECT failed: %s (line %d)\n", #cond, __LINE__); return 1; }        \
    } while (0)

static uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// smooth blobs + noise, random alpha (the path ignores it: rgb_to_lab.wgsl:78)
static std::vector<uint8_t> make_image(uint32_t w, uint32_t h, uint64_t seed)
{
    std::vector<uint8_t> img((size_t)w * h * 4);
    uint64_t s = seed;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const uint64_t r = splitmix(s);
            uint8_t *p = &img[((size_t)y * w + x) * 4];
            p[0] = (uint8_t)((x * 255u / w + (r & 31u)) & 255u);
            p[1] = (uint8_t)((y * 255u / h + ((r >> 8) & 31u)) & 255u);
            p[2] = (uint8_t)((((x / 64u) * 37u + (y / 64u) * 91u) + ((r >> 16) & 15u)) & 255u);
            p[3] = (uint8_t)(r >> 24);
        }
    return img;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "nogpu")) {
        kmg_group *g = nullptr;
        const int rc = kmg_group_create(nullptr, &g);
        if (rc == KMG_OK) { printf("created\n"); kmg_group_destroy(g); return 0; }
        printf("error %d %s\n", rc, kmg_last_error());
        return (rc == KMG_ERR_NO_DEVICE && strlen(kmg_last_error()) > 0) ? 0 : 1;
    }
    if (argc != 6 || strcmp(argv[1], "run")) { fprintf(stderr, "usage\n"); return 2; }
    const uint32_t ranks = (uint32_t)atoi(argv[2]), k = (uint32_t)atoi(argv[3]), w = (uint32_t)atoi(argv[4]), h = (uint32_t)atoi(argv[5]);
    const uint64_t n = (uint64_t)w * h;
    const std::vector<uint8_t> img = make_image(w, h, 0xC0FFEEull + w);

    for (uint32_t shrink : {256u, 0u}) {
        // ---- reference: one processor ----
        kmg_options po;
        kmg_default_options(&po);
        po.device = 0;
        po.shrink_max_dim = shrink;
        // (the TEST's own switch: the library reads no environment variable for this -- kmg_options.strategy)
        if (const char *e = getenv("CHECK_GROUP_STRATEGY")) po.strategy = !strcmp(e, "table") ? KMG_STRATEGY_TABLE : (!strcmp(e, "scan") ? KMG_STRATEGY_SCAN : KMG_STRATEGY_AUTO);
        kmg_processor *p = nullptr;
        CHECK(kmg_processor_create_ex(&po, &p));
        kmg_group_options go;
        kmg_default_group_options(&go);
        go.n_devices = ranks;
        for (uint32_t i = 0; i < ranks; ++i) go.devices[i] = 0;
        go.flags = ranks == 1 ? KMG_GROUP_FORCE_COLLECTIVES : KMG_GROUP_LOOPBACK;
        go.processor = po;
        kmg_group *g = nullptr;
        CHECK(kmg_group_create(&go, &g));
        uint32_t n_local = 0, first = 9, world = 0;
        int version = -1;
        CHECK(kmg_group_info(g, &n_local, &first, &world, &version));
        EXPECT(n_local == ranks && first == 0 && world == ranks);
        EXPECT(ranks > 1 ? version == 0 : version > 0);
        printf("group of %u rank(s), shrink %u, rccl version %d\n", ranks, shrink, version);

        // host-buffer calls: palette / find / reduce, every mode
        std::vector<uint8_t> a(n * 4), b(n * 4);
        std::vector<uint8_t> pa(k * 4), pb(k * 4);
        uint32_t ca = 0, cb = 0;
        for (int algo : {KMG_ALGO_KMEANS, KMG_ALGO_OCTREE}) {
            CHECK(kmg_palette(p, img.data(), w, h, k, algo, pa.data(), &ca));
            CHECK(kmg_group_palette(g, img.data(), w, h, k, algo, pb.data(), &cb));
            EXPECT(ca == cb && !memcmp(pa.data(), pb.data(), ca * 4));
        }
        for (int mode : {KMG_MODE_REPLACE, KMG_MODE_DITHER, KMG_MODE_MELD}) {
            CHECK(kmg_find(p, img.data(), w, h, pa.data(), ca, mode, a.data()));
            CHECK(kmg_group_find(g, img.data(), w, h, pa.data(), ca, mode, b.data()));
            EXPECT(a == b);
            CHECK(kmg_reduce(p, img.data(), w, h, k, KMG_ALGO_KMEANS, mode, a.data()));
            CHECK(kmg_group_reduce(g, img.data(), w, h, k, KMG_ALGO_KMEANS, mode, b.data()));
            EXPECT(a == b);
        }
        CHECK(kmg_reduce(p, img.data(), w, h, k, KMG_ALGO_OCTREE, KMG_MODE_DITHER, a.data()));
        CHECK(kmg_group_reduce(g, img.data(), w, h, k, KMG_ALGO_OCTREE, KMG_MODE_DITHER, b.data()));
        EXPECT(a == b);
        // a band without rows: more ranks than image rows
        if (ranks > 1) {
            CHECK(kmg_find(p, img.data(), w, 1, pa.data(), ca, KMG_MODE_DITHER, a.data()));
            CHECK(kmg_group_find(g, img.data(), w, 1, pa.data(), ca, KMG_MODE_DITHER, b.data()));
            EXPECT(!memcmp(a.data(), b.data(), (size_t)w * 4));
        }
        // a batch of whole images per device (BASELINE config 4 as placed): three images, every one = kmg_reduce
        {
            const uint32_t bw[3] = {w, w / 2, 97}, bh[3] = {h / 3, h / 2, 61};
            std::vector<uint8_t> out[3], want[3];
            const uint8_t *src[3];
            uint8_t *dst[3];
            for (int i = 0; i < 3; ++i) {
                out[i].resize((size_t)bw[i] * bh[i] * 4); want[i].resize(out[i].size());
                src[i] = img.data() + 4 * i;                      // (any readable pixels)
                dst[i] = out[i].data();
                CHECK(kmg_reduce(p, src[i], bw[i], bh[i], k, KMG_ALGO_KMEANS, KMG_MODE_DITHER, want[i].data()));
            }
            CHECK(kmg_group_reduce_batch(g, 3, src, bw, bh, k, KMG_ALGO_KMEANS, KMG_MODE_DITHER, dst));
            for (int i = 0; i < 3; ++i) EXPECT(out[i] == want[i]);
        }
        // argument errors of the group layer: refused before any rank starts, the group stays usable
        if (shrink == 0) {
            kmg_group_options bad = go;
            if (ranks == 1) {
                bad.n_devices = 2; bad.devices[0] = 0; bad.devices[1] = 0; bad.flags = 0;      // one device twice without the loopback exchange
                kmg_group *gb = nullptr;
                EXPECT(kmg_group_create(&bad, &gb) == KMG_ERR_INVALID_ARGUMENT && gb == nullptr && strstr(kmg_last_error(), "twice"));
            }
            bad = go; bad.devices[0] = 4096;
            kmg_group *gb = nullptr;
            EXPECT(kmg_group_create(&bad, &gb) == KMG_ERR_NO_DEVICE && gb == nullptr);
            kmg_group_lloyd *gl = nullptr;
            CHECK(kmg_group_lloyd_create(g, k, &gl));
            uint32_t it = 0;
            EXPECT(kmg_group_lloyd_run(gl, &it) == KMG_ERR_INVALID_ARGUMENT);                    // no bands yet
            EXPECT(kmg_group_lloyd_step(gl) == KMG_ERR_INVALID_ARGUMENT);
            std::vector<const uint8_t *> nb(ranks, nullptr);
            std::vector<uint32_t> r0(ranks, 0u), rs(ranks, 0u);
            rs[0] = h + 1;                                                                       // a band that leaves the image
            nb[0] = img.data();
            EXPECT(kmg_group_lloyd_bind(gl, nb.data(), r0.data(), rs.data(), w, h, nullptr, 0u) == KMG_ERR_INVALID_ARGUMENT);
            rs[0] = 1; nb[0] = nullptr;                                                          // rows without pixels
            EXPECT(kmg_group_lloyd_bind(gl, nb.data(), r0.data(), rs.data(), w, h, nullptr, 0u) == KMG_ERR_INVALID_ARGUMENT);
            kmg_group_lloyd_destroy(gl);
            EXPECT(kmg_group_lloyd_create(g, 0, &gl) == KMG_ERR_INVALID_ARGUMENT);
            EXPECT(kmg_group_processor(g, ranks) == nullptr && kmg_group_processor(g, 0) != nullptr && kmg_group_stream(g, 0) != nullptr);
        }
        // errors reach the caller with a message
        EXPECT(kmg_group_reduce(g, img.data(), w, h, 0, KMG_ALGO_KMEANS, KMG_MODE_REPLACE, b.data()) == KMG_ERR_INVALID_ARGUMENT);
        EXPECT(kmg_group_find(g, img.data(), 0, h, pa.data(), ca, KMG_MODE_REPLACE, b.data()) == KMG_ERR_INVALID_ARGUMENT && strlen(kmg_last_error()) > 0);

        // ---- device-pointer level: the sharded loop against kmg_lloyd_run (full resolution) ----
        if (shrink == 0) {
            HIPCHECK(hipSetDevice(0));
            uint8_t *d_img = nullptr;
            uint32_t *d_lab_one = nullptr, *d_lab_group = nullptr;
            HIPCHECK(hipMalloc((void **)&d_img, n * 4));
            HIPCHECK(hipMalloc((void **)&d_lab_one, n * 4));
            HIPCHECK(hipMalloc((void **)&d_lab_group, n * 4));
            HIPCHECK(hipMemcpy(d_img, img.data(), n * 4, hipMemcpyHostToDevice));
            kmg_lloyd *s = nullptr;
            CHECK(kmg_lloyd_create(p, k, &s));
            CHECK(kmg_lloyd_init_centroids(s, d_img, w, h, nullptr));
            std::vector<float> c_init(4 * k), c_one(4 * k), c_group(4 * k);
            CHECK(kmg_lloyd_get_centroids(s, c_init.data(), nullptr));
            uint32_t it_one = 0, it_group = 0;
            CHECK(kmg_lloyd_run(s, d_img, n, d_lab_one, &it_one, nullptr));
            CHECK(kmg_lloyd_get_centroids(s, c_one.data(), nullptr));
            std::vector<uint32_t> lab_one(n), lab_group(n);
            HIPCHECK(hipMemcpy(lab_one.data(), d_lab_one, n * 4, hipMemcpyDeviceToHost));

            std::vector<const uint8_t *> bands(ranks);
            std::vector<uint32_t *> labs(ranks);
            std::vector<uint32_t> row0(ranks), rows(ranks);
            for (uint32_t i = 0; i < ranks; ++i) {
                row0[i] = (uint32_t)((uint64_t)i * h / ranks);
                rows[i] = (uint32_t)((uint64_t)(i + 1) * h / ranks) - row0[i];
                bands[i] = d_img + (size_t)row0[i] * w * 4;
                labs[i] = d_lab_group + (size_t)row0[i] * w;
            }
            for (uint32_t flags : {0u, (uint32_t)KMG_GROUP_OVERLAP, (uint32_t)KMG_GROUP_CELLS}) {
                if ((flags & KMG_GROUP_CELLS) && k > 256) continue;
                kmg_group_lloyd *gl = nullptr;
                CHECK(kmg_group_lloyd_create(g, k, &gl));
                CHECK(kmg_group_lloyd_bind(gl, bands.data(), row0.data(), rows.data(), w, h, labs.data(), flags));
                CHECK(kmg_group_lloyd_init(gl));
                CHECK(kmg_group_lloyd_get_centroids(gl, c_group.data()));
                EXPECT(!memcmp(c_init.data(), c_group.data(), sizeof(float) * 4 * k));
                HIPCHECK(hipMemset(d_lab_group, 0xFF, n * 4));
                CHECK(kmg_group_lloyd_run(gl, &it_group));
                CHECK(kmg_group_lloyd_get_centroids(gl, c_group.data()));
                HIPCHECK(hipMemcpy(lab_group.data(), d_lab_group, n * 4, hipMemcpyDeviceToHost));
                EXPECT(it_group == it_one);
                EXPECT(!memcmp(c_one.data(), c_group.data(), sizeof(float) * 4 * k));
                EXPECT(lab_one == lab_group);
                // step by step: prime + steps = the same centroids as the loop after the same number of updates
                CHECK(kmg_group_lloyd_set_centroids(gl, c_init.data()));
                CHECK(kmg_group_lloyd_prime(gl));
                for (uint32_t i = 0; i <= it_one; ++i) CHECK(kmg_group_lloyd_step(gl));
                CHECK(kmg_group_lloyd_sync(gl));
                CHECK(kmg_group_lloyd_get_centroids(gl, c_group.data()));
                EXPECT(!memcmp(c_one.data(), c_group.data(), sizeof(float) * 4 * k));
                HIPCHECK(hipMemcpy(lab_group.data(), d_lab_group, n * 4, hipMemcpyDeviceToHost));
                EXPECT(lab_one == lab_group);
                if (flags == (uint32_t)KMG_GROUP_CELLS) {
                    // the fused form of the cell-sharded loop (the cube pass adds into the accumulators, the band's label pass updates
                    // from the all-reduced sums and clears them): prime + n steps = n + 1 updates of the plain loop
                    CHECK(kmg_group_lloyd_bind(gl, bands.data(), row0.data(), rows.data(), w, h, labs.data(), KMG_GROUP_CELLS | KMG_GROUP_FUSED_UPDATE));
                    CHECK(kmg_group_lloyd_set_centroids(gl, c_init.data()));
                    CHECK(kmg_group_lloyd_prime(gl));
                    for (uint32_t i = 0; i < it_one; ++i) CHECK(kmg_group_lloyd_step(gl));
                    CHECK(kmg_group_lloyd_sync(gl));
                    CHECK(kmg_group_lloyd_get_centroids(gl, c_group.data()));
                    EXPECT(!memcmp(c_one.data(), c_group.data(), sizeof(float) * 4 * k));
                    EXPECT(kmg_group_lloyd_run(gl, &it_group) == KMG_ERR_INVALID_ARGUMENT && strstr(kmg_last_error(), "FUSED"));
                    CHECK(kmg_group_lloyd_bind(gl, bands.data(), row0.data(), rows.data(), w, h, labs.data(), flags));
                }
                if (flags == 0u) {
                    // the loop reads the convergence count between update and re-assignment: the fused form is for _prime / _step
                    CHECK(kmg_group_lloyd_bind(gl, bands.data(), row0.data(), rows.data(), w, h, labs.data(), KMG_GROUP_FUSED_UPDATE));
                    EXPECT(kmg_group_lloyd_run(gl, &it_group) == KMG_ERR_INVALID_ARGUMENT && strstr(kmg_last_error(), "FUSED"));
                    CHECK(kmg_group_lloyd_bind(gl, bands.data(), row0.data(), rows.data(), w, h, labs.data(), flags));
                }
                int strategy = -1;
                EXPECT(kmg_group_lloyd_member(gl, 0, &strategy) != nullptr);
                printf("flags %u: %u iterations, strategy of rank 0: %s\n", flags, it_group, strategy ? "table" : "scan");
                kmg_group_lloyd_destroy(gl);
            }
            // ---- a BATCH of three images, each tiled over all ranks (BASELINE config 4 as north_star words it): one accumulator block, one
            // all-reduce per iteration, per-image convergence = three single-processor runs, byte for byte ----
            {
                const uint32_t bw[3] = {w, w / 2 + 3, 300}, bh[3] = {h, h / 2 + 1, 2 * ranks + 1};     // (the third: bands of two or three rows)
                const size_t off_px[3] = {0, 5, 11};                                                  // (any readable pixels of the image)
                std::vector<std::vector<float>> c_want(3, std::vector<float>(4 * k)), c0(3, std::vector<float>(4 * k));
                std::vector<std::vector<uint32_t>> lab_want(3);
                uint32_t it_want[3] = {0, 0, 0};
                std::vector<const uint8_t *> b_px(3 * ranks);
                std::vector<uint32_t *> b_lab(3 * ranks);
                std::vector<uint32_t> b_row0(3 * ranks), b_rows(3 * ranks);
                size_t lab_off = 0, lab_total = 0;
                for (int im = 0; im < 3; ++im) lab_total += (size_t)bw[im] * bh[im];
                uint32_t *d_lab_batch = nullptr;                                                      // the three label maps, one after the other
                HIPCHECK(hipMalloc((void **)&d_lab_batch, lab_total * 4));
                for (int im = 0; im < 3; ++im) {
                    const uint8_t *d_im = d_img + off_px[im] * 4;
                    const uint64_t nn = (uint64_t)bw[im] * bh[im];
                    CHECK(kmg_lloyd_init_centroids(s, d_im, bw[im], bh[im], nullptr));
                    CHECK(kmg_lloyd_get_centroids(s, c0[im].data(), nullptr));
                    CHECK(kmg_lloyd_run(s, d_im, nn, d_lab_one, &it_want[im], nullptr));
                    CHECK(kmg_lloyd_get_centroids(s, c_want[im].data(), nullptr));
                    lab_want[im].resize(nn);
                    HIPCHECK(hipMemcpy(lab_want[im].data(), d_lab_one, nn * 4, hipMemcpyDeviceToHost));
                    for (uint32_t i = 0; i < ranks; ++i) {
                        const uint32_t a = (uint32_t)((uint64_t)i * bh[im] / ranks), b = (uint32_t)((uint64_t)(i + 1) * bh[im] / ranks);
                        b_row0[im * ranks + i] = a; b_rows[im * ranks + i] = b - a;
                        b_px[im * ranks + i] = d_im + (size_t)a * bw[im] * 4;
                        b_lab[im * ranks + i] = d_lab_batch + lab_off + (size_t)a * bw[im];
                    }
                    lab_off += nn;
                }
                EXPECT(lab_off == lab_total);
                kmg_group_lloyd *gb = nullptr;
                CHECK(kmg_group_lloyd_create_batch(g, k, 3, &gb));
                EXPECT(kmg_group_lloyd_bind_batch(gb, b_px.data(), b_row0.data(), b_rows.data(), bw, bh, b_lab.data(), KMG_GROUP_CELLS) == KMG_ERR_INVALID_ARGUMENT);
                uint32_t it3[3] = {9, 9, 9};
                EXPECT(kmg_group_lloyd_run(gb, it3) == KMG_ERR_INVALID_ARGUMENT);                    // (no bands yet)
                CHECK(kmg_group_lloyd_bind_batch(gb, b_px.data(), b_row0.data(), b_rows.data(), bw, bh, b_lab.data(), 0u));
                EXPECT(kmg_group_lloyd_run(gb, it3) == KMG_ERR_INVALID_ARGUMENT && strstr(kmg_last_error(), "run_batch"));
                CHECK(kmg_group_lloyd_init(gb));                                                     // the sharded initialisation, batched collectives
                for (uint32_t im = 0; im < 3; ++im) {
                    CHECK(kmg_group_lloyd_get_centroids_image(gb, im, c_group.data()));
                    EXPECT(!memcmp(c0[im].data(), c_group.data(), sizeof(float) * 4 * k));
                }
                HIPCHECK(hipMemset(d_lab_batch, 0xFF, lab_total * 4));
                CHECK(kmg_group_lloyd_run_batch(gb, it3));
                lab_off = 0;
                for (uint32_t im = 0; im < 3; ++im) {
                    const uint64_t nn = (uint64_t)bw[im] * bh[im];
                    CHECK(kmg_group_lloyd_get_centroids_image(gb, im, c_group.data()));
                    EXPECT(it3[im] == it_want[im]);
                    EXPECT(!memcmp(c_want[im].data(), c_group.data(), sizeof(float) * 4 * k));
                    std::vector<uint32_t> got(nn);
                    HIPCHECK(hipMemcpy(got.data(), d_lab_batch + lab_off, nn * 4, hipMemcpyDeviceToHost));
                    EXPECT(got == lab_want[im]);
                    lab_off += nn;
                }
                // set_centroids_image + prime + steps: the same centroids as the loop after the same number of updates (image 1)
                for (uint32_t im = 0; im < 3; ++im) CHECK(kmg_group_lloyd_set_centroids_image(gb, im, c0[im].data()));
                CHECK(kmg_group_lloyd_prime(gb));
                for (uint32_t i = 0; i <= it_want[1]; ++i) CHECK(kmg_group_lloyd_step(gb));
                CHECK(kmg_group_lloyd_sync(gb));
                CHECK(kmg_group_lloyd_get_centroids_image(gb, 1, c_group.data()));
                EXPECT(!memcmp(c_want[1].data(), c_group.data(), sizeof(float) * 4 * k));
                printf("batch of 3 tiled images: iterations %u %u %u\n", it3[0], it3[1], it3[2]);
                kmg_group_lloyd_destroy(gb);
                HIPCHECK(hipFree(d_lab_batch));
            }
            kmg_lloyd_destroy(s);
            HIPCHECK(hipFree(d_img)); HIPCHECK(hipFree(d_lab_one)); HIPCHECK(hipFree(d_lab_group));
        }
        kmg_group_destroy(g);
        kmg_processor_destroy(p);
    }
    printf("ok group of %u\n", ranks);
    return 0;
}
