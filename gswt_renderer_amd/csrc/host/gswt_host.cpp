// gswt_host.cpp -- libgswt_host.so: C++17 mirror of the reference's host-side hot-path code
// (scene::Scene loader, wangtile::WangTile worker, host half of GSWTRenderer::render) behind the
// C ABI of include/gswt_host.h.  file:line citations are into zengyf131/gswt_renderer.
//
// Float discipline: f32, one rounding per operator, the reference's operand order (build with
// -ffp-contract=off).  Third-party behaviour restated without the crate sources (rand 0.9.2
// StdRng / random_range, petgraph 0.8.3 toposort) is "parity unpinned" -- see DESIGN.md.
#include "../../../include/gswt_host.h"

#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <deque>
#include <fstream>
#include <list>
#include <map>
#include <memory>
#include <regex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "gswt_math.h"
#include "gswt_surface.h"

using namespace gswt_host;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// No C++ exception may unwind through the C ABI (the callers are ctypes / Rust FFI): every extern "C" body that returns a
// status is a function-try-block closed by this handler.
#define GSWT_CATCH(NAME)                                                                                    \
    catch (const std::bad_alloc&) { return fail(GSWT_ERR_CAPACITY, NAME ": out of memory"); }                \
    catch (const std::exception& e) { return fail(GSWT_ERR_IO, NAME ": %s", e.what()); }                     \
    catch (...) { return fail(GSWT_ERR_IO, NAME ": unknown exception"); }

// ------------------------------------------------------------------------------------------
// halves (utils.rs:66-73; half 2.7.1 f16::from_f32, round to nearest even)
// ------------------------------------------------------------------------------------------
uint32_t float_to_half(float value)
{
    uint32_t x;
    memcpy(&x, &value, 4);
    uint32_t sign = x & 0x80000000u, exp = x & 0x7F800000u, man = x & 0x007FFFFFu;
    if (exp == 0x7F800000u) {
        uint32_t nan_bit = man == 0 ? 0 : 0x0200u;
        return (sign >> 16) | 0x7C00u | nan_bit | (man >> 13);
    }
    uint32_t half_sign = sign >> 16;
    int32_t half_exp = (int32_t)(exp >> 23) - 127 + 15;
    if (half_exp >= 0x1F) return half_sign | 0x7C00u;
    if (half_exp <= 0) {
        if (14 - half_exp > 24) return half_sign;
        man |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - half_exp);
        uint32_t half_man = man >> shift;
        uint32_t round_bit = 1u << (shift - 1);
        if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0) half_man += 1;
        return half_sign | half_man;
    }
    uint32_t half_e = (uint32_t)half_exp << 10, half_man = man >> 13;
    const uint32_t round_bit = 0x00001000u;
    if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0) return (half_sign | half_e | half_man) + 1;
    return half_sign | half_e | half_man;
}

uint32_t pack_half_2x16(float x, float y) { return float_to_half(x) | (float_to_half(y) << 16); }

// ------------------------------------------------------------------------------------------
// scene::Scene (scene.rs:50-57): splat_count + 32 B/splat buffer
// ------------------------------------------------------------------------------------------
struct Scene {
    size_t splat_count = 0;
    std::vector<uint8_t> buffer;
    const float* f(size_t i) const { return reinterpret_cast<const float*>(buffer.data() + 32 * i); }
    float* f(size_t i) { return reinterpret_cast<float*>(buffer.data() + 32 * i); }
};

// Scene::load, scene.rs:115-212
void scene_load(Scene& sc, const float* verts62, size_t n)
{
    const float SH_C0 = 0.28209479177387814f;
    std::vector<float> size_list(n);
    std::vector<uint32_t> size_index(n);
    for (size_t i = 0; i < n; i++) {
        const float* s = verts62 + 62 * i;
        float size = std::exp(s[55]) * std::exp(s[56]) * std::exp(s[57]);
        float opacity = 1.0f / (1.0f + std::exp(-s[54]));
        size_list[i] = size * opacity;
        size_index[i] = (uint32_t)i;
    }
    std::stable_sort(size_index.begin(), size_index.end(), [&](uint32_t a, uint32_t b) { return size_list[b] < size_list[a]; });
    sc.splat_count = n;
    sc.buffer.assign(32 * n, 0);
    for (size_t i = 0; i < n; i++) {
        const float* s = verts62 + 62 * (size_t)size_index[i];
        uint8_t* row = sc.buffer.data() + 32 * i;
        float fv[6] = {s[0], s[1], s[2], std::exp(s[55]), std::exp(s[56]), std::exp(s[57])};
        memcpy(row, fv, 24);
        row[24] = rust_as_u8((0.5f + SH_C0 * s[6]) * 255.0f);
        row[25] = rust_as_u8((0.5f + SH_C0 * s[7]) * 255.0f);
        row[26] = rust_as_u8((0.5f + SH_C0 * s[8]) * 255.0f);
        row[27] = rust_as_u8((1.0f / (1.0f + std::exp(-s[54]))) * 255.0f);
        float qlen = std::sqrt(((s[58] * s[58] + s[59] * s[59]) + s[60] * s[60]) + s[61] * s[61]);
        for (int k = 0; k < 4; k++) row[28 + k] = rust_as_u8(((s[58 + k] / qlen) + 1.0f) * 0.5f * 255.0f);
    }
}

// Scene::parse_file_header, scene.rs:72-112
int parse_ply_header(const uint8_t* data, size_t len, size_t* header_size, size_t* count)
{
    size_t pos = 0;
    size_t splat_count = 0;
    for (int i = 0; i <= 65; i++) {
        const uint8_t* nl = (const uint8_t*)memchr(data + pos, '\n', len - pos);
        if (!nl) break;
        std::string line((const char*)data + pos, (size_t)(nl - (data + pos)) + 1);
        pos = (size_t)(nl - data) + 1;
        if (line == "end_header\n") { *header_size = pos; *count = splat_count; return GSWT_OK; }
        if (line.rfind("element vertex ", 0) == 0) {
            char* endp = nullptr;
            std::string num = line.substr(15, line.size() - 16);
            errno = 0;
            unsigned long long v = strtoull(num.c_str(), &endp, 10);
            if (num.empty() || *endp != '\0' || errno == ERANGE || num[0] == '-' || num[0] == '+')
                return fail(GSWT_ERR_IO, "Scene::parse_file_header(): bad vertex count '%s'", num.c_str());
            splat_count = (size_t)v;
        }
    }
    return fail(GSWT_ERR_IO, "Scene::parse_file_header(): ERROR: the file is not correctly formatted.");
}

// Scene::generate_texture, scene.rs:306-411
void generate_texture(const uint8_t* rows32, size_t n, uint32_t* tex)
{
    for (size_t i = 0; i < n; i++) {
        const uint8_t* row = rows32 + 32 * i;
        float fb[6];
        memcpy(fb, row, 24);
        uint32_t* t = tex + 8 * i;
        memcpy(t, row, 12);
        t[3] = 0;
        memcpy(&t[7], row + 24, 4);
        float rot[4];
        for (int k = 0; k < 4; k++) rot[k] = ((float)row[28 + k] / 255.0f) * 2.0f - 1.0f;
        float r[9];
        r[0] = 1.0f - 2.0f * (rot[2] * rot[2] + rot[3] * rot[3]);
        r[1] = 2.0f * (rot[1] * rot[2] + rot[0] * rot[3]);
        r[2] = 2.0f * (rot[1] * rot[3] - rot[0] * rot[2]);
        r[3] = 2.0f * (rot[1] * rot[2] - rot[0] * rot[3]);
        r[4] = 1.0f - 2.0f * (rot[1] * rot[1] + rot[3] * rot[3]);
        r[5] = 2.0f * (rot[2] * rot[3] + rot[0] * rot[1]);
        r[6] = 2.0f * (rot[1] * rot[3] + rot[0] * rot[2]);
        r[7] = 2.0f * (rot[2] * rot[3] - rot[0] * rot[1]);
        r[8] = 1.0f - 2.0f * (rot[1] * rot[1] + rot[2] * rot[2]);
        float m[9];   // r * diag(scale): the products with the zeros of diag() contribute +-0
        for (int c = 0; c < 3; c++)
            for (int rr = 0; rr < 3; rr++) {
                float acc = 0.0f;
                for (int k = 0; k < 3; k++) {
                    float sk = (k == c) ? fb[3 + c] : 0.0f;
                    float term = r[3 * k + rr] * sk;
                    acc = (k == 0) ? term : acc + term;
                }
                m[3 * c + rr] = acc;
            }
        float sg[6];
        sg[0] = m[0] * m[0] + m[3] * m[3] + m[6] * m[6];
        sg[1] = m[0] * m[1] + m[3] * m[4] + m[6] * m[7];
        sg[2] = m[0] * m[2] + m[3] * m[5] + m[6] * m[8];
        sg[3] = m[1] * m[1] + m[4] * m[4] + m[7] * m[7];
        sg[4] = m[1] * m[2] + m[4] * m[5] + m[7] * m[8];
        sg[5] = m[2] * m[2] + m[5] * m[5] + m[8] * m[8];
        t[4] = pack_half_2x16(4.0f * sg[0], 4.0f * sg[1]);
        t[5] = pack_half_2x16(4.0f * sg[2], 4.0f * sg[3]);
        t[6] = pack_half_2x16(4.0f * sg[4], 4.0f * sg[5]);
    }
}

// raw depth of Scene::sort_self, scene.rs:537-552
void raw_depth(const Scene& sc, const float* vp, std::vector<int32_t>& out)
{
    out.resize(sc.splat_count);
    for (size_t i = 0; i < sc.splat_count; i++) {
        const float* p = sc.f(i);
        out[i] = rust_as_i32((vp[2] * p[0] + vp[6] * p[1] + vp[10] * p[2]) * 4096.0f);
    }
}

// Scene::sort_raw_depth_vec, scene.rs:655-698; order_out[j] = index into the concatenation
void sort_raw_depth(const int32_t* depths, size_t n, uint32_t* order_out)
{
    if (n == 0) return;
    int32_t mn = depths[0], mx = depths[0];
    for (size_t i = 1; i < n; i++) { mn = std::min(mn, depths[i]); mx = std::max(mx, depths[i]); }
    const int32_t size16 = 65536;
    float depth_inv = (float)(size16 - 1) / (float)(int32_t)(mx - mn);
    std::vector<uint32_t> counts(size16, 0), starts(size16, 0);
    std::vector<int32_t> bucket(n);
    for (size_t i = 0; i < n; i++) {
        int32_t d = rust_as_i32(std::floor((float)(int32_t)(depths[i] - mn) * depth_inv));
        d = std::min(std::max(d, 0), size16 - 1);
        bucket[i] = d;
        counts[d]++;
    }
    for (int32_t i = 1; i < size16; i++) starts[i] = starts[i - 1] + counts[i - 1];
    for (size_t i = 0; i < n; i++) order_out[starts[bucket[i]]++] = (uint32_t)i;
    std::reverse(order_out, order_out + n);
}

// ------------------------------------------------------------------------------------------
// zip container (stored / deflate) for load_scene_zip, scene.rs:1030-1141
// ------------------------------------------------------------------------------------------
struct ZipEntry { std::string name; uint32_t method, comp_size, size, local_off; };

uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int zip_list(const uint8_t* z, size_t len, std::vector<ZipEntry>& out)
{
    if (len < 22) return fail(GSWT_ERR_IO, "zip: too short");
    size_t eocd = SIZE_MAX;
    for (size_t i = len - 22;; i--) {
        if (rd32(z + i) == 0x06054b50u) { eocd = i; break; }
        if (i == 0 || len - i > 22 + 65535) break;
    }
    if (eocd == SIZE_MAX) return fail(GSWT_ERR_IO, "zip: end of central directory not found");
    uint32_t n = rd16(z + eocd + 10), cd_off = rd32(z + eocd + 16);
    size_t p = cd_off;
    for (uint32_t i = 0; i < n; i++) {
        if (p + 46 > len || rd32(z + p) != 0x02014b50u) return fail(GSWT_ERR_IO, "zip: bad central directory");
        ZipEntry e;
        e.method = rd16(z + p + 10);
        e.comp_size = rd32(z + p + 20);
        e.size = rd32(z + p + 24);
        uint16_t nl = rd16(z + p + 28), xl = rd16(z + p + 30), cl = rd16(z + p + 32);
        e.local_off = rd32(z + p + 42);
        if (p + 46 + nl > len) return fail(GSWT_ERR_IO, "zip: truncated entry name");
        e.name.assign((const char*)z + p + 46, nl);
        out.push_back(e);
        p += 46 + (size_t)nl + xl + cl;
    }
    return GSWT_OK;
}

int zip_read(const uint8_t* z, size_t len, const ZipEntry& e, std::vector<uint8_t>& out)
{
    size_t p = e.local_off;
    if (p + 30 > len || rd32(z + p) != 0x04034b50u) return fail(GSWT_ERR_IO, "zip: bad local header for %s", e.name.c_str());
    size_t data = p + 30 + rd16(z + p + 26) + rd16(z + p + 28);
    if (data + e.comp_size > len) return fail(GSWT_ERR_IO, "zip: truncated data for %s", e.name.c_str());
    out.resize(e.size);
    if (e.method == 0) {
        if (e.comp_size != e.size) return fail(GSWT_ERR_IO, "zip: stored size mismatch for %s", e.name.c_str());
        memcpy(out.data(), z + data, e.size);
        return GSWT_OK;
    }
    if (e.method != 8) return fail(GSWT_ERR_IO, "zip: unsupported compression method %u for %s", e.method, e.name.c_str());
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return fail(GSWT_ERR_IO, "zip: inflateInit2 failed");
    zs.next_in = const_cast<Bytef*>(z + data);
    zs.avail_in = e.comp_size;
    zs.next_out = out.data();
    zs.avail_out = e.size;
    int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.total_out != e.size) return fail(GSWT_ERR_IO, "zip: inflate failed for %s", e.name.c_str());
    return GSWT_OK;
}

// ------------------------------------------------------------------------------------------
// rand 0.9 StdRng restatement: ChaCha12, key from rand_core's PCG32 seed expansion ("unpinned")
// ------------------------------------------------------------------------------------------
struct StdRng {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[16];
    int pos = 16;
    explicit StdRng(uint64_t seed = 0) { reseed(seed); }
    void reseed(uint64_t state)
    {
        for (int i = 0; i < 8; i++) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
        counter = 0;
        pos = 16;
    }
    static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    void block()
    {
        uint32_t st[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u};
        for (int i = 0; i < 8; i++) st[4 + i] = key[i];
        st[12] = (uint32_t)counter; st[13] = (uint32_t)(counter >> 32); st[14] = 0; st[15] = 0;
        uint32_t w[16];
        memcpy(w, st, sizeof(w));
        auto qr = [&](int a, int b, int c, int d) {
            w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 16);
            w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 12);
            w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 8);
            w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 7);
        };
        for (int r = 0; r < 6; r++) {
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) buf[i] = w[i] + st[i];
        counter++;
        pos = 0;
    }
    uint32_t next_u32() { if (pos >= 16) block(); return buf[pos++]; }
    uint32_t random_range_u32(uint32_t n)        // random_range(0..n), 32-bit usize (wasm32)
    {
        uint64_t m = (uint64_t)next_u32() * n;
        uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
        if (lo > (uint32_t)(0u - n)) {
            uint32_t hi2 = (uint32_t)(((uint64_t)next_u32() * n) >> 32);
            if ((uint64_t)lo + hi2 > 0xFFFFFFFFull) hi += 1;
        }
        return hi;
    }
    float random_range_f32_inclusive(float low, float high)
    {
        const float max_rand = 1.0f - 1.1920928955078125e-07f;
        float scale = (high - low) / max_rand;
        while (scale * max_rand + low > high) scale = std::nextafter(scale, -INFINITY);
        uint32_t bits = (next_u32() >> 9) | 0x3F800000u;
        float v12;
        memcpy(&v12, &bits, 4);
        float v01 = v12 - 1.0f;
        return v01 * scale + low;
    }
};

// ------------------------------------------------------------------------------------------
// petgraph DiGraph restatement ("unpinned"): adjacency newest-edge-first, swap_remove on nodes
// ------------------------------------------------------------------------------------------
struct DiGraph {
    std::vector<size_t> weights;
    std::vector<std::vector<int>> out, inc;
    std::vector<std::pair<int, int>> edges;   // (-1,-1) when removed
    int add_node(size_t w) { weights.push_back(w); out.emplace_back(); inc.emplace_back(); return (int)weights.size() - 1; }
    void add_edge(int a, int b)
    {
        edges.emplace_back(a, b);
        int e = (int)edges.size() - 1;
        out[a].insert(out[a].begin(), e);
        inc[b].insert(inc[b].begin(), e);
    }
    void remove_node(int n)
    {
        std::vector<int> es = out[n];
        es.insert(es.end(), inc[n].begin(), inc[n].end());
        for (int e : es) {
            if (edges[e].first < 0) continue;
            int a = edges[e].first, b = edges[e].second;
            out[a].erase(std::find(out[a].begin(), out[a].end(), e));
            inc[b].erase(std::find(inc[b].begin(), inc[b].end(), e));
            edges[e] = {-1, -1};
        }
        int last = (int)weights.size() - 1;
        if (n != last) {
            weights[n] = weights[last];
            out[n] = out[last];
            inc[n] = inc[last];
            for (int e : out[n]) edges[e].first = n;
            for (int e : inc[n]) edges[e].second = n;
        }
        weights.pop_back(); out.pop_back(); inc.pop_back();
    }
    // petgraph::algo::toposort; returns true and `order`, or false and `cycle_node`
    bool toposort(std::vector<int>& order, int& cycle_node) const
    {
        const int n = (int)weights.size();
        std::vector<char> discovered(n, 0), finished(n, 0);
        std::vector<int> finish_stack, stack;
        for (int i = n - 1; i >= 0; i--) {
            if (discovered[i]) continue;
            stack.push_back(i);
            while (!stack.empty()) {
                int nx = stack.back();
                if (!discovered[nx]) {
                    discovered[nx] = 1;
                    for (int e : out[nx]) {
                        int succ = edges[e].second;
                        if (succ == nx) { cycle_node = nx; return false; }
                        if (!discovered[succ]) stack.push_back(succ);
                    }
                } else {
                    stack.pop_back();
                    if (!finished[nx]) { finished[nx] = 1; finish_stack.push_back(nx); }
                }
            }
        }
        std::reverse(finish_stack.begin(), finish_stack.end());
        std::fill(discovered.begin(), discovered.end(), 0);
        for (int i : finish_stack) {
            stack.clear();
            stack.push_back(i);
            bool cycle = false;
            while (!stack.empty()) {
                int node = stack.back();
                stack.pop_back();
                if (discovered[node]) continue;
                discovered[node] = 1;
                for (int e : inc[node]) {
                    int pred = edges[e].first;
                    if (!discovered[pred]) stack.push_back(pred);
                }
                if (cycle) { cycle_node = node; return false; }
                cycle = true;
            }
        }
        order = finish_stack;
        return true;
    }
};

// ------------------------------------------------------------------------------------------
// structure.rs data contracts
// ------------------------------------------------------------------------------------------
enum { SORT_DISTANCE = 0, SORT_VIEWPORT = 1, SORT_OBJECT = 2, SORT_GRAPH = 3 };
enum { MERGE_NONE = 0, MERGE_AXIS = 1, MERGE_EDGE = 2 };
enum { SURFACE_NONE = 0, SURFACE_HEIGHTMAP = 1, SURFACE_SPHERE = 2 };
enum { HMAP_TEXTURE = 0, HMAP_RANDOM = 1, HMAP_SLOPEX = 2, HMAP_SLOPEY = 3, HMAP_DUALSLOPE = 4 };
enum { TR_NONE = 0, TR_SPAWNING = 1, TR_CHANGING_HIGHER = 2, TR_CHANGING_LOWER = 3 };   // Changing(false) / Changing(true)
enum { MS_NONE = 0, MS_FROM = 1, MS_TO = 2 };

struct TileBaseData {   // structure.rs:545-554
    size_t splat_count = 0;
    std::vector<int32_t> raw_depth;
    std::vector<uint32_t> gs_index, gs_lod_id;
};

struct Neighbor { bool some = false; int x = 0, y = 0, slot = 0; };

struct Corner { V3 pos; M3 to_world; };
struct Edge { V3 pos; V3 normal; };

struct TileInstance {   // structure.rs:495-509
    size_t lod = 0, tile = 0, view_id = 0;
    V3 tile_offset;
    size_t map_index = 0;
    int mx = 0, my = 0;
    V3 tile_center;
    int merge = MS_NONE;
    std::vector<size_t> merged_from;
    size_t merged_to = 0;
    int transition = TR_NONE;
    float spawning = 0.0f;
    M3 to_local;
    bool has_corner = false;
    Corner corner[4];
    Edge edge[4];
};

struct RenderDataValue {   // structure.rs:686-694
    size_t splat_count = 0;
    std::vector<uint32_t> gs_index, gs_map_id, gs_lod_id;
    std::vector<size_t> merge_from_vec;
    int32_t single_lod_id = -1;
    bool has_lod = false;
};

int status_hash(int tr) { return tr; }   // TileTransitionStatusHash: Spawning carries no payload

}  // namespace

struct gswt_tileset {
    int n_lod = 0, n_tile = 0;
    std::vector<std::vector<Scene>> s;
};

struct gswt_wang {
    gswt_user_data user{};
    std::vector<float> height_tex_copy;
    std::vector<std::vector<Scene>> tiles;
    size_t n_lod = 0, n_tile = 0, n_view = 0;
    bool initialized = false;
    int map_w = 0, map_h = 0;
    std::vector<std::unique_ptr<TileInstance>> tile_map;    // [x * h + y]
    std::vector<Neighbor> neighbor_map;                     // [(x * h + y) * 4 + slot]
    int center_x = 0, center_y = 0;
    V3 camera_pos;
    std::vector<V3> presort_dirs;
    StdRng rng{0};
    // preprocess outputs
    std::vector<uint32_t> tex;                              // tile_splats_merged.tex_data
    size_t merged_count = 0;
    std::vector<std::vector<uint32_t>> merge_offset;        // [lod][tile]
    std::vector<float> lod_avg_scale;
    std::vector<V3> tile_center, aabb_lo, aabb_hi;          // per tile id
    std::vector<TileBaseData> base;                         // [(lod * n_tile + tile) * n_view + view]
    std::vector<gswt_base_list> base_lists;
    // configure outputs
    std::vector<float> height_map;
    int hm_w = 0, hm_h = 0;
    std::vector<float> lod_transition_dist;
    // LRU cache of merged lists (wangtile.rs:37,426-427,575-593,672-675)
    std::list<std::pair<std::string, RenderDataValue>> lru;
    std::unordered_map<std::string, std::list<std::pair<std::string, RenderDataValue>>::iterator> lru_index;
    // last sort result
    std::vector<gswt_sorted_tile> sorted;
    std::vector<uint32_t> m_gs, m_map, m_lod;
    std::vector<gswt_merge_group> m_groups;
    std::vector<gswt_merge_member> m_members;
    bool device_merge = false;
    std::vector<const int32_t*> rd_ptrs;
    std::vector<uint32_t> rd_counts, rd_offsets;
    std::vector<float> wk_center, wk_aabb, wk_dirs;          // flat tables of gswt_wang_worker_config
    std::vector<uint32_t> wk_counts;
    std::vector<int32_t> wk_nb;

    TileBaseData& tb(size_t l, size_t t, size_t v) { return base[(l * n_tile + t) * n_view + v]; }
    const TileBaseData& tb(size_t l, size_t t, size_t v) const { return base[(l * n_tile + t) * n_view + v]; }
    TileInstance* at(int x, int y) { return tile_map[(size_t)x * map_h + y].get(); }
    const TileInstance* at(int x, int y) const { return tile_map[(size_t)x * map_h + y].get(); }
    const Neighbor& nb(int x, int y, int slot) const { return neighbor_map[((size_t)x * map_h + y) * 4 + slot]; }
    size_t map_to_index(int x, int y) const { return (size_t)x * map_h + y; }
    void index_to_map(size_t idx, int& x, int& y) const { x = (int)(idx / map_h); y = (int)(idx % map_h); }
    V3 coord_to_pos(int cx, int cy) const { return {(float)cx * user.tile_width, (float)cy * user.tile_width, 0.0f}; }
    void map_to_coord(int x, int y, int& cx, int& cy) const
    {
        cx = x + center_x - (int)user.tile_map_half_wh[0];
        cy = y + center_y - (int)user.tile_map_half_wh[1];
    }
};

namespace {

SurfaceParams surface_params(const gswt_wang& w)
{
    SurfaceParams p;
    p.map_w = w.map_w; p.map_h = w.map_h; p.half_w = (int)w.user.tile_map_half_wh[0]; p.half_h = (int)w.user.tile_map_half_wh[1];
    p.center_x = w.center_x; p.center_y = w.center_y;
    p.tile_width = w.user.tile_width; p.surface_type = (int)w.user.surface_type;
    for (int k = 0; k < 3; k++) p.hm_scale[k] = w.user.height_map_scale[k];
    p.sphere_radius = w.user.sphere_radius;
    p.height_map = w.height_map.empty() ? nullptr : w.height_map.data(); p.hm_w = w.hm_w; p.hm_h = w.hm_h;
    return p;
}

// ---- WangTile::preprocess, wangtile.rs:71-255 ---------------------------------------------
int preprocess(gswt_wang& w)
{
    w.n_lod = w.tiles.size();
    w.n_tile = w.tiles[0].size();
    for (auto& lod : w.tiles) {
        if (lod.size() != w.n_tile) return fail(GSWT_ERR_BAD_ARG, "WangTile::preprocess: ragged tile set");
        for (auto& sc : lod)
            if (sc.splat_count == 0) return fail(GSWT_ERR_BAD_ARG, "WangTile::preprocess: empty tile scene");
    }
    if (w.n_lod > 16) return fail(GSWT_ERR_BAD_ARG, "WangTile::preprocess: more than 16 LODs");
    w.tile_center.resize(w.n_tile); w.aabb_lo.resize(w.n_tile); w.aabb_hi.resize(w.n_tile);
    for (size_t t = 0; t < w.n_tile; t++) {
        Scene& s0 = w.tiles[0][t];
        V3 lo, hi, avg;                                     // compute_aabb_and_center, scene.rs:830-861
        for (size_t i = 0; i < s0.splat_count; i++) {
            const float* p = s0.f(i);
            V3 pos{p[0], p[1], p[2]};
            avg = avg + pos;
            if (i == 0) { lo = pos; hi = pos; }
            else {
                lo = {std::min(lo.x, pos.x), std::min(lo.y, pos.y), std::min(lo.z, pos.z)};
                hi = {std::max(hi.x, pos.x), std::max(hi.y, pos.y), std::max(hi.z, pos.z)};
            }
        }
        avg = avg / (float)s0.splat_count;
        for (size_t l = 0; l < w.n_lod; l++) {              // height normalisation, :84-87
            Scene& sc = w.tiles[l][t];
            for (size_t i = 0; i < sc.splat_count; i++) {
                float* p = sc.f(i);
                p[0] += 0.0f; p[1] += 0.0f; p[2] += -avg.z;
            }
        }
        lo.z -= avg.z; hi.z -= avg.z; avg.z = 0.0f;
        w.aabb_lo[t] = lo; w.aabb_hi[t] = hi;
        w.tile_center[t] = avg / (float)w.n_lod;            // :106-107
    }
    // merge, :113-125
    w.merge_offset.assign(w.n_lod, std::vector<uint32_t>(w.n_tile, 0));
    size_t total = 0;
    for (size_t l = 0; l < w.n_lod; l++)
        for (size_t t = 0; t < w.n_tile; t++) { w.merge_offset[l][t] = (uint32_t)total; total += w.tiles[l][t].splat_count; }
    w.merged_count = total;
    w.tex.assign(8 * total, 0);
    for (size_t l = 0; l < w.n_lod; l++)
        for (size_t t = 0; t < w.n_tile; t++)
            generate_texture(w.tiles[l][t].buffer.data(), w.tiles[l][t].splat_count, w.tex.data() + 8 * (size_t)w.merge_offset[l][t]);
    // avg scale, :128-142
    w.lod_avg_scale.clear();
    for (size_t l = 0; l < w.n_lod; l++) {
        float sum = 0.0f;
        size_t num = 0;
        for (size_t t = 0; t < w.n_tile; t++) {
            const Scene& sc = w.tiles[l][t];
            float ssum = 0.0f;                               // compute_scale_sum, scene.rs:863-873
            for (size_t i = 0; i < sc.splat_count; i++) { const float* p = sc.f(i); ssum += p[3]; ssum += p[4]; ssum += p[5]; }
            sum += ssum;
            num += sc.splat_count * 3;
        }
        float avg = sum / (float)num;
        if (l > 0 && !(avg > w.lod_avg_scale[l - 1]))
            return fail(GSWT_ERR_BAD_ARG, "WangTile::preprocess: assertion failed: avg_scale > self.lod_avg_scale[l - 1] (lod %zu)", l);
        w.lod_avg_scale.push_back(avg);
    }
    // presort views, :144-174
    const float raw[9][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {1, 0, -1}, {-1, 0, -1}, {0, 1, -1}, {0, -1, -1}, {0, 0, -1}};
    float proj[16];
    perspective(90.0f, 1.0f, 0.1f, 10.0f, proj);
    w.presort_dirs.clear();
    std::vector<std::vector<float>> vps;
    for (int k = 0; k < 9; k++) {
        V3 d = normalize(V3{raw[k][0], raw[k][1], raw[k][2]});
        w.presort_dirs.push_back(d);
        float view[16], vp[16];
        V3 up = (d.x != 0.0f || d.y != 0.0f) ? V3{0, 0, 1} : V3{0, 1, 0};
        look_at_rh(V3{0, 0, 0}, d, up, view);
        mat4_mul(proj, view, vp);
        vps.emplace_back(vp, vp + 16);
    }
    w.n_view = 9;
    w.base.assign(w.n_lod * w.n_tile * w.n_view, TileBaseData{});
    for (size_t l = 0; l < w.n_lod; l++)
        for (size_t t = 0; t < w.n_tile; t++)
            for (size_t k = 0; k < w.n_view; k++) raw_depth(w.tiles[l][t], vps[k].data(), w.tb(l, t, k).raw_depth);
    // base lists, :221-252
    std::vector<int32_t> cat;
    std::vector<uint32_t> order;
    for (size_t l = 0; l < w.n_lod; l++)
        for (size_t t = 0; t < w.n_tile; t++)
            for (size_t k = 0; k < w.n_view; k++) {
                const std::vector<int32_t>& a = w.tb(l, t, k).raw_depth;
                cat.assign(a.begin(), a.end());
                size_t n0 = a.size();
                if (l + 1 < w.n_lod) { const auto& b = w.tb(l + 1, t, k).raw_depth; cat.insert(cat.end(), b.begin(), b.end()); }
                order.resize(cat.size());
                sort_raw_depth(cat.data(), cat.size(), order.data());
                TileBaseData& tb = w.tb(l, t, k);
                tb.splat_count = cat.size();
                tb.gs_index.resize(cat.size()); tb.gs_lod_id.resize(cat.size());
                for (size_t j = 0; j < order.size(); j++) {
                    uint32_t o = order[j];
                    if (o < n0) { tb.gs_index[j] = o + w.merge_offset[l][t]; tb.gs_lod_id[j] = (uint32_t)l; }
                    else { tb.gs_index[j] = (o - (uint32_t)n0) + w.merge_offset[l + 1][t]; tb.gs_lod_id[j] = (uint32_t)l + 1; }
                }
            }
    w.base_lists.resize(w.base.size());
    for (size_t i = 0; i < w.base.size(); i++) {
        w.base_lists[i].gs_index = w.base[i].gs_index.data();
        w.base_lists[i].gs_lod_id = w.base[i].gs_lod_id.data();
        w.base_lists[i].splat_count = (uint32_t)w.base[i].splat_count;
        w.base_lists[i]._pad = 0;
    }
    return GSWT_OK;
}

// ---- topology, wangtile.rs:257-338 -------------------------------------------------------
// slots 0 = W, 1 = N, 2 = E, 3 = S; each entry = (neighbour map coord, the slot THIS tile occupies for it).
// Sphere: the map is an unfolded icosahedron strip of 5 x 2 square blocks (block_w = map_w / 5 = map_h / 2);
// block edges wrap onto other blocks, some with a quarter turn (slot numbers change).
void compute_map_neighbors(const gswt_wang& w, int x, int y, Neighbor out[4])
{
    for (int k = 0; k < 4; k++) out[k] = Neighbor{};
    if (w.user.surface_type == SURFACE_SPHERE) {
        const int map_w = w.map_w, map_h = w.map_h, block_w = map_w / 5;
        const int bidx = 5 * x / map_w, bidy = 2 * y / map_h;
        const int bx = x - bidx * block_w, by = y - bidy * block_w;
        if (bx > 0) out[0] = {true, x - 1, y, 2};
        else if (bidy == 0) out[0] = {true, (map_w + x - 1) % map_w, y + block_w, 2};
        else out[0] = {true, (map_w + x - by - 1) % map_w, map_h - 1, 1};
        if (bx < block_w - 1) out[2] = {true, x + 1, y, 0};
        else if (bidy == 0) out[2] = {true, (x + block_w - by) % map_w, 0, 3};
        else out[2] = {true, (x + 1) % map_w, y - block_w, 0};
        if (y > 0) out[3] = {true, x, y - 1, 1};
        else out[3] = {true, (map_w + bidx * block_w - 1) % map_w, block_w - 1 - bx, 2};
        if (y < map_h - 1) out[1] = {true, x, y + 1, 3};
        else out[1] = {true, (bidx * block_w + block_w) % map_w, 2 * block_w - 1 - bx, 0};
        return;
    }
    if (x > 0) out[0] = {true, x - 1, y, 2};                 // west
    if (x < w.map_w - 1) out[2] = {true, x + 1, y, 0};       // east
    if (y > 0) out[3] = {true, x, y - 1, 1};                 // south
    if (y < w.map_h - 1) out[1] = {true, x, y + 1, 3};       // north
}

// ---- height map helpers, wangtile.rs:1220-1349 ---------------------------------------------
void cubic_weight(float t, float wgt[4])
{
    wgt[0] = ((-0.5f * t + 1.0f) * t - 0.5f) * t;
    wgt[1] = ((1.5f * t - 2.5f) * t) * t + 1.0f;
    wgt[2] = ((-1.5f * t + 2.0f) * t + 0.5f) * t;
    wgt[3] = ((0.5f * t - 0.5f) * t) * t;
}

std::vector<float> map_resize(const std::vector<float>& src, int fw, int fh, int tw, int th)
{
    std::vector<float> out((size_t)tw * th);
    for (int j = 0; j < th; j++)
        for (int i = 0; i < tw; i++) {
            float ux = (float)i / (float)tw, uy = (float)j / (float)th;
            float x = ux * (float)fw - 0.5f, y = uy * (float)fh - 0.5f;
            long x0 = (long)std::floor(x), y0 = (long)std::floor(y);
            float dx = x - (float)x0, dy = y - (float)y0;
            float wx[4], wy[4];
            cubic_weight(dx, wx);
            cubic_weight(dy, wy);
            float result = 0.0f;
            for (int jj = 0; jj < 4; jj++)
                for (int ii = 0; ii < 4; ii++) {
                    float val = hm_texel(src.data(), fw, fh, x0 + ii - 1, y0 + jj - 1);
                    result += val * wx[ii] * wy[jj];
                }
            out[(size_t)j * tw + i] = result;
        }
    return out;
}

void surface_mapping(const gswt_wang& w, int mcx, int mcy, V3 pos, bool to_world, V3& new_pos, M3& transform)
{
    surface_mapping(surface_params(w), mcx, mcy, pos, to_world, new_pos, transform);
}

// ---- compute_corner_edge, wangtile.rs:1609-1669 -------------------------------------------
bool compute_corner_edge(const gswt_wang& w, int x, int y, float tile_center_z, Corner corner[4], Edge edge[4])
{
    if (w.user.tile_sort_type != SORT_GRAPH && w.user.merge_type != MERGE_EDGE) return false;
    const int dco[4][2] = {{0, 0}, {0, 1}, {1, 1}, {1, 0}};
    for (int ci = 0; ci < 4; ci++) {
        bool done = false;
        const Neighbor& n1 = w.nb(x, y, ci);
        if (n1.some) {
            const TileInstance* ni = w.at(n1.x, n1.y);
            if (ni && ni->has_corner) { corner[ci] = ni->corner[(n1.slot + 1) % 4]; done = true; }
        }
        if (!done) {
            const Neighbor& n2 = w.nb(x, y, (ci + 3) % 4);
            if (n2.some) {
                const TileInstance* ni = w.at(n2.x, n2.y);
                if (ni && ni->has_corner) { corner[ci] = ni->corner[n2.slot]; done = true; }
            }
        }
        if (!done) {
            int cx, cy;
            w.map_to_coord(x + dco[ci][0], y + dco[ci][1], cx, cy);
            V3 cpos = w.coord_to_pos(cx, cy) + V3{0, 0, 1} * tile_center_z;
            surface_mapping(w, x, y, cpos, true, corner[ci].pos, corner[ci].to_world);
        }
    }
    for (int ei = 0; ei < 4; ei++) {
        const Corner& c1 = corner[ei];
        const Corner& c2 = corner[(ei + 1) % 4];
        V3 epos = (c1.pos + c2.pos) / 2.0f;
        V3 cdir = c2.pos - c1.pos;
        V3 n1 = c1.to_world * V3{0, 0, 1}, n2 = c2.to_world * V3{0, 0, 1};
        V3 normal = (n1 + n2) / 2.0f;
        edge[ei].pos = epos;
        edge[ei].normal = normalize(cross(normal, cdir));
    }
    return true;
}

// ---- lod_select_spatial + update_lod, wangtile.rs:1496-1607 -------------------------------
void lod_select_spatial(const gswt_wang& w, int x, int y, V3 cam, size_t& lod_out, int& status_out)
{
    int cx, cy;
    w.map_to_coord(x, y, cx, cy);
    V3 pos_offset = w.coord_to_pos(cx, cy);
    const TileInstance* ti = w.at(x, y);
    const std::vector<float>& D = w.lod_transition_dist;
    float center_dist = distance(ti->tile_center, cam);
    size_t sel = D.size() - 1;
    for (size_t l = 0; l < D.size(); l++)
        if (center_dist <= D[l]) { sel = l; break; }
    int status = TR_NONE;
    if (w.user.lod_blending) {
        V3 lo = w.aabb_lo[ti->tile], hi = w.aabb_hi[ti->tile];
        V3 pts[8];
        int npts = 0;
        if (w.user.lod_bbox_check) {
            pts[0] = lo; pts[1] = {lo.x, lo.y, hi.z}; pts[2] = {lo.x, hi.y, lo.z}; pts[3] = {lo.x, hi.y, hi.z};
            pts[4] = {hi.x, lo.y, lo.z}; pts[5] = {hi.x, lo.y, hi.z}; pts[6] = {hi.x, hi.y, lo.z}; pts[7] = hi;
            npts = 8;
        } else { pts[0] = w.tile_center[ti->tile]; npts = 1; }
        float mn = -1.0f, mx = -1.0f;
        for (int k = 0; k < npts; k++) {
            V3 q; M3 tr;
            surface_mapping(w, x, y, pts[k] + pos_offset, true, q, tr);
            float d = distance(q, cam);
            if (mn < 0.0f || d < mn) mn = d;
            if (mx < 0.0f || d > mx) mx = d;
        }
        const float r = w.user.lod_transition_width_ratio, tol = w.user.lod_dist_tolerance;
        if (sel > 0 && mn < D[sel - 1] * (1.0f + r) + tol) status = TR_CHANGING_HIGHER;
        if (sel < D.size() - 1 && mx > D[sel] * (1.0f - r) - tol) status = TR_CHANGING_LOWER;
    }
    lod_out = sel;
    status_out = status;
}

void update_lod(gswt_wang& w, V3 cam)
{
    V3 cc = w.coord_to_pos(w.center_x, w.center_y);
    float cam_u = (cam.x - cc.x) / w.user.tile_width;
    float cam_v = (cam.y - cc.y) / w.user.tile_width;
    for (int i = 0; i < w.map_w; i++)
        for (int j = 0; j < w.map_h; j++) {
            size_t lod; int st;
            lod_select_spatial(w, i, j, cam, lod, st);
            TileInstance* ti = w.at(i, j);
            ti->lod = lod;
            ti->transition = st;
            ti->spawning = 0.0f;
            if (w.user.lod_blending && w.user.surface_type != SURFACE_SPHERE) {
                float bf = 1.0f;
                if (i == 0) bf *= 1.0f - cam_u;
                else if (i == w.map_w - 1) bf *= cam_u;
                if (j == 0) bf *= 1.0f - cam_v;
                else if (j == w.map_h - 1) bf *= cam_v;
                if (bf != 1.0f) { ti->transition = TR_SPAWNING; ti->spawning = bf; }
            }
        }
}

// ---- update_tile_map, wangtile.rs:1671-1781 ------------------------------------------------
void tile_id_to_color(size_t tid, int c[4])
{
    c[0] = (int)(tid % 16 / 8 % 2); c[1] = (int)(tid % 16 / 4 % 2); c[2] = (int)(tid % 16 / 2 % 2); c[3] = (int)(tid % 16 % 2);
}

void update_tile_map(gswt_wang& w, V3 cam)
{
    w.camera_pos = cam;
    if (w.user.surface_type == SURFACE_SPHERE) { w.center_x = 0; w.center_y = 0; }      // the sphere map never shifts, :1721-1723
    else {
    int prev_cx = w.center_x, prev_cy = w.center_y;
    w.center_x = (int)std::floor(cam.x / w.user.tile_width);
    w.center_y = (int)std::floor(cam.y / w.user.tile_width);
    std::vector<std::unique_ptr<TileInstance>> new_map((size_t)w.map_w * w.map_h);
    for (int i = 0; i < w.map_w; i++)
        for (int j = 0; j < w.map_h; j++) {
            int px = i + w.center_x - prev_cx, py = j + w.center_y - prev_cy;
            if (px >= 0 && px < w.map_w && py >= 0 && py < w.map_h) {
                const TileInstance* prev = w.at(px, py);
                if (prev) {
                    auto ti = std::make_unique<TileInstance>(*prev);
                    ti->lod = 0; ti->view_id = 0;
                    ti->map_index = w.map_to_index(i, j);
                    ti->mx = i; ti->my = j;
                    ti->merge = MS_NONE; ti->merged_from.clear(); ti->merged_to = 0;
                    ti->transition = TR_NONE; ti->spawning = 0.0f;
                    new_map[(size_t)i * w.map_h + j] = std::move(ti);
                }
            }
        }
    w.tile_map = std::move(new_map);
    }
    for (int i = 0; i < w.map_w; i++)
        for (int j = 0; j < w.map_h; j++) {
            if (w.at(i, j)) continue;
            int cx, cy;
            w.map_to_coord(i, j, cx, cy);
            V3 tile_offset = w.coord_to_pos(cx, cy);
            int color[4] = {0, 0, 0, 0};
            for (int idx = 0; idx < 4; idx++) {
                const Neighbor& n = w.nb(i, j, idx);
                if (n.some) {
                    const TileInstance* nt = w.at(n.x, n.y);
                    if (nt) { int nc[4]; tile_id_to_color(nt->tile, nc); color[idx] = nc[n.slot]; }
                    else color[idx] = (int)w.rng.random_range_u32(2);
                } else color[idx] = (int)w.rng.random_range_u32(2);
            }
            uint32_t center_option = w.rng.random_range_u32(w.user.center_option);
            size_t tile_id = (size_t)(color[0] * 8 + color[1] * 4 + color[2] * 2 + color[3]) + 16 * (size_t)center_option;
            auto ti = std::make_unique<TileInstance>();
            ti->lod = 0; ti->tile = tile_id; ti->view_id = 0;
            ti->tile_offset = tile_offset;
            ti->map_index = w.map_to_index(i, j);
            ti->mx = i; ti->my = j;
            V3 base_center = w.tile_center[tile_id];
            surface_mapping(w, i, j, base_center + tile_offset, false, ti->tile_center, ti->to_local);
            ti->has_corner = compute_corner_edge(w, i, j, base_center.z, ti->corner, ti->edge);
            w.tile_map[(size_t)i * w.map_h + j] = std::move(ti);
        }
    update_lod(w, cam);
}

// ---- selective merging, wangtile.rs:720-1027 -----------------------------------------------
void set_merge_none(TileInstance* t) { t->merge = MS_NONE; t->merged_from.clear(); t->merged_to = 0; }

void selective_merge_edge(gswt_wang& w, V3 cam, const float* vp)
{
    struct E { size_t mi; int ei; float dabs, ndot; };
    std::vector<E> edges;
    std::vector<char> check((size_t)w.map_w * w.map_h, 0);
    for (int i = 0; i < w.map_w; i++)
        for (int j = 0; j < w.map_h; j++) {
            size_t mi = w.map_to_index(i, j);
            check[mi] = 1;
            TileInstance* ti = w.at(i, j);
            set_merge_none(ti);
            for (int n_i = 0; n_i < 4; n_i++) {
                const Neighbor& n = w.nb(i, j, n_i);
                if (!n.some) continue;
                if (check[w.map_to_index(n.x, n.y)]) continue;
                const Edge& e = ti->edge[n_i];
                const Corner& c1 = ti->corner[n_i];
                const Corner& c2 = ti->corner[(n_i + 1) % 4];
                V3 vd = e.pos - cam;
                float vlen = magnitude(vd);
                if (is_zero(vd)) continue;
                if (dot(vd, c1.to_world.col(2)) > 0.0f || dot(vd, c2.to_world.col(2)) > 0.0f) continue;
                float a4[4] = {c1.pos.x, c1.pos.y, c1.pos.z, 1.0f}, b4[4] = {c2.pos.x, c2.pos.y, c2.pos.z, 1.0f}, p1[4], p2[4];
                mat4_vec(vp, a4, p1);
                mat4_vec(vp, b4, p2);
                V3 q1 = V3{p1[0], p1[1], p1[2]} / p1[3], q2 = V3{p2[0], p2[1], p2[2]} / p2[3];
                const float clip = 1.0f;
                auto outside = [&](V3 p) { return p.z < -clip || p.x < -clip || p.x > clip || p.y < -clip || p.y > clip; };
                if (outside(q1) && outside(q2)) continue;
                float dabs = std::fabs(dot(e.normal, vd));
                edges.push_back({mi, n_i, dabs, dabs / vlen});
            }
        }
    std::stable_sort(edges.begin(), edges.end(), [](const E& a, const E& b) { return a.dabs < b.dabs; });
    size_t topk = 0;
    std::vector<int> merge_map((size_t)w.map_w * w.map_h, -1);
    std::vector<std::vector<size_t>> groups;
    for (const E& e : edges) {
        if (topk >= w.user.merge_topk) break;
        if (e.ndot > w.user.merge_dot_threshold) continue;
        int x, y;
        w.index_to_map(e.mi, x, y);
        const Neighbor& n = w.nb(x, y, e.ei);
        size_t ni = w.map_to_index(n.x, n.y);
        int a = merge_map[e.mi], b = merge_map[ni];
        if (a < 0 && b < 0) { groups.push_back({e.mi, ni}); merge_map[e.mi] = merge_map[ni] = (int)groups.size() - 1; }
        else if (a >= 0 && b < 0) { groups[a].push_back(ni); merge_map[ni] = a; }
        else if (a < 0 && b >= 0) { groups[b].push_back(e.mi); merge_map[e.mi] = b; }
        else if (a != b) {
            for (size_t g : groups[b]) merge_map[g] = a;
            groups[a].insert(groups[a].end(), groups[b].begin(), groups[b].end());
            groups[b].clear();
        }
        topk++;
    }
    for (size_t i = 0; i < groups.size(); i++) {                 // fix non-convex groups, :959-990
        std::unordered_set<size_t> seen;
        size_t j = 0;
        while (j < groups[i].size()) {
            int tx, ty;
            w.index_to_map(groups[i][j], tx, ty);
            for (int s = 0; s < 4; s++) {
                const Neighbor& n = w.nb(tx, ty, s);
                if (!n.some) continue;
                size_t nidx = w.map_to_index(n.x, n.y);
                if (std::find(groups[i].begin(), groups[i].end(), nidx) != groups[i].end()) continue;
                if (!seen.insert(nidx).second) {
                    int other = merge_map[nidx];
                    if (other >= 0) {
                        for (size_t g : groups[other]) merge_map[g] = (int)i;
                        std::vector<size_t> tmp;
                        tmp.swap(groups[other]);
                        groups[i].insert(groups[i].end(), tmp.begin(), tmp.end());
                    } else { groups[i].push_back(nidx); merge_map[nidx] = (int)i; }
                }
            }
            j++;
        }
    }
    for (auto& grp : groups) {
        if (grp.empty()) continue;
        std::vector<size_t> g = grp;
        std::sort(g.begin(), g.end());
        float mind = 3.402823466e+38f;
        size_t mini = 0;
        for (size_t k = 0; k < g.size(); k++) {
            int x, y;
            w.index_to_map(g[k], x, y);
            float d2 = distance2(w.at(x, y)->tile_center, cam);
            if (mind > d2) { mind = d2; mini = k; }
        }
        for (size_t k = 0; k < g.size(); k++)
            if (k != mini) {
                int x, y;
                w.index_to_map(g[k], x, y);
                TileInstance* t = w.at(x, y);
                t->merge = MS_TO; t->merged_to = g[mini]; t->merged_from.clear();
            }
        int x, y;
        w.index_to_map(g[mini], x, y);
        TileInstance* t = w.at(x, y);
        t->merge = MS_FROM; t->merged_from = g;
    }
}

void selective_merge_axis(gswt_wang& w, V3 cam, const float* vp)
{
    int cmx = w.center_x - w.center_x + (int)w.user.tile_map_half_wh[0];   // coord_to_map(center_coord)
    int cmy = w.center_y - w.center_y + (int)w.user.tile_map_half_wh[1];
    if (w.user.surface_type == SURFACE_SPHERE) {                            // nearest not-MergedTo tile, :725-740
        float min_dist = -1.0f;
        cmx = cmy = 0;
        for (size_t idx = 0; idx < (size_t)w.map_w * w.map_h; idx++) {
            int x, y;
            w.index_to_map(idx, x, y);
            const TileInstance* t = w.at(x, y);
            if (t->merge == MS_TO) continue;
            const V3 dv = cam - t->tile_center;
            const float d2 = dot(dv, dv);
            if (min_dist < 0.0f || d2 < min_dist) { min_dist = d2; cmx = x; cmy = y; }
        }
    }
    float best = 0.0f;
    int merge_dir = -1;
    V3 cam_dir = normalize(V3{vp[2], vp[6], vp[10]});
    for (int ci = 0; ci < 4; ci++) {
        const Neighbor& n = w.nb(cmx, cmy, ci);
        if (!n.some) continue;
        float dp = dot(normalize(w.at(n.x, n.y)->tile_center - cam), cam_dir);
        if (best < dp) { best = dp; merge_dir = ci; }
    }
    if (merge_dir < 0) return;
    const int mn[4][2] = {{3, 1}, {0, 2}, {1, 3}, {2, 0}};
    int mx = cmx, my = cmy;
    for (int i = 0; i < w.user.merge_tile_dist[0]; i++) {
        const Neighbor& n = w.nb(mx, my, merge_dir);
        if (!n.some) return;        // reference: unwrap() panic when the map is too small
        mx = n.x; my = n.y;
    }
    for (int i = w.user.merge_tile_dist[0]; i < w.user.merge_tile_dist[1]; i++) {
        const Neighbor& a = w.nb(mx, my, mn[merge_dir][0]);
        const Neighbor& b = w.nb(mx, my, mn[merge_dir][1]);
        const Neighbor& f = w.nb(mx, my, merge_dir);
        if (!a.some || !b.some) return;
        size_t cidx = w.map_to_index(mx, my);
        TileInstance *tc = w.at(mx, my), *ta = w.at(a.x, a.y), *tbb = w.at(b.x, b.y);
        if (tc->merge != MS_NONE || ta->merge != MS_NONE || tbb->merge != MS_NONE) break;
        tc->merge = MS_FROM; tc->merged_from = {w.map_to_index(a.x, a.y), cidx, w.map_to_index(b.x, b.y)};
        ta->merge = MS_TO; ta->merged_to = cidx;
        tbb->merge = MS_TO; tbb->merged_to = cidx;
        if (!f.some) return;
        mx = f.x; my = f.y;
    }
}

// ---- tile orderings, wangtile.rs:1029-1218 -------------------------------------------------
std::vector<size_t> sort_by_key_desc(std::vector<std::pair<size_t, float>>& sv)
{
    std::stable_sort(sv.begin(), sv.end(), [](const auto& a, const auto& b) { return a.second < b.second; });
    std::reverse(sv.begin(), sv.end());
    std::vector<size_t> out;
    for (auto& e : sv) out.push_back(e.first);
    return out;
}

std::vector<size_t> sort_tiles_object_pos(const gswt_wang& w, V3 cam)
{
    std::vector<std::pair<size_t, float>> sv;
    for (size_t idx = 0; idx < (size_t)w.map_w * w.map_h; idx++) {
        const TileInstance* ti = w.tile_map[idx].get();
        if (ti->merge == MS_TO) continue;
        sv.emplace_back(idx, distance2(cam, ti->tile_center));
    }
    return sort_by_key_desc(sv);
}

std::vector<size_t> sort_tiles_object_vp(const gswt_wang& w, const float* vp)
{
    std::vector<std::pair<size_t, float>> sv;
    for (size_t idx = 0; idx < (size_t)w.map_w * w.map_h; idx++) {
        const TileInstance* ti = w.tile_map[idx].get();
        if (ti->merge == MS_TO) continue;
        V3 p = ti->tile_center;
        sv.emplace_back(idx, (vp[2] * p.x + vp[6] * p.y) + vp[10] * p.z);
    }
    return sort_by_key_desc(sv);
}

std::vector<size_t> sort_tiles_object_bfs(const gswt_wang& w, V3 cam)
{
    int sx = 0, sy = 0;
    float min_d = -1.0f;
    for (size_t idx = 0; idx < (size_t)w.map_w * w.map_h; idx++) {
        const TileInstance* ti = w.tile_map[idx].get();
        if (ti->merge == MS_TO) continue;
        float d = distance2(cam, ti->tile_center);
        if (min_d < 0.0f || d < min_d) { min_d = d; w.index_to_map(idx, sx, sy); }
    }
    std::vector<char> check((size_t)w.map_w * w.map_h, 0);
    std::vector<size_t> out;
    std::deque<std::pair<int, int>> queue;
    queue.emplace_back(sx, sy);
    check[w.map_to_index(sx, sy)] = 1;
    while (!queue.empty()) {
        auto [x, y] = queue.front();
        queue.pop_front();
        out.push_back(w.map_to_index(x, y));
        for (int s = 0; s < 4; s++) {
            const Neighbor& n = w.nb(x, y, s);
            if (n.some && !check[w.map_to_index(n.x, n.y)]) { queue.emplace_back(n.x, n.y); check[w.map_to_index(n.x, n.y)] = 1; }
        }
    }
    std::reverse(out.begin(), out.end());
    return out;
}

std::vector<size_t> sort_tiles_object_graph(const gswt_wang& w, V3 cam)
{
    DiGraph g;
    std::vector<int> node_map((size_t)w.map_w * w.map_h, -1);
    for (int i = 0; i < w.map_w; i++)
        for (int j = 0; j < w.map_h; j++)
            if (w.at(i, j)->merge != MS_TO) node_map[w.map_to_index(i, j)] = g.add_node(w.map_to_index(i, j));
    auto node_of = [&](int x, int y) {
        const TileInstance* ti = w.at(x, y);
        if (ti->merge == MS_TO) return node_map[ti->merged_to];
        return node_map[w.map_to_index(x, y)];
    };
    std::vector<char> check((size_t)w.map_w * w.map_h, 0);
    for (int i = 0; i < w.map_w; i++)
        for (int j = 0; j < w.map_h; j++) {
            const TileInstance* ti = w.at(i, j);
            int this_node = node_of(i, j);
            check[w.map_to_index(i, j)] = 1;
            for (int s = 0; s < 4; s++) {
                const Neighbor& n = w.nb(i, j, s);
                if (!n.some || check[w.map_to_index(n.x, n.y)]) continue;
                int nnode = node_of(n.x, n.y);
                if (this_node == nnode) continue;
                V3 vd = ti->edge[s].pos - cam;
                if (is_zero(vd)) continue;
                float dr = dot(ti->edge[s].normal, vd);
                if (dr > 0.0f) g.add_edge(this_node, nnode);
                else if (dr < 0.0f) g.add_edge(nnode, this_node);
            }
        }
    std::vector<size_t> out, removed;
    for (;;) {
        std::vector<int> order;
        int cyc = -1;
        if (g.toposort(order, cyc)) {
            for (int node : order)
                if (!g.inc[node].empty() || !g.out[node].empty()) out.push_back(g.weights[node]);
            break;
        }
        removed.push_back(g.weights[cyc]);
        g.remove_node(cyc);
    }
    out.insert(out.end(), removed.begin(), removed.end());
    std::reverse(out.begin(), out.end());
    return out;
}

size_t choose_presort_view(const gswt_wang& w, const M3& transform, V3 pos, V3 cam)
{
    V3 dl = transform * normalize(pos - cam);
    size_t best = 0;
    float best_err = 1000.0f;
    for (size_t i = 0; i < w.presort_dirs.size(); i++) {
        V3 pd = w.presort_dirs[i];
        float ex = dl.x - pd.x, ey = dl.y - pd.y, ez = dl.z - pd.z;
        float err = (ex * ex + ey * ey) + ez * ez;
        if (err < best_err) { best = i; best_err = err; }
    }
    return best;
}

// merged-group list, wangtile.rs:595-670
void build_merged_value(const gswt_wang& w, const std::vector<size_t>& from_vec, size_t view_id, size_t head_lod, RenderDataValue& val)
{
    bool do_transition = false;
    for (size_t mi : from_vec)
        if (w.tile_map[mi]->transition != TR_NONE) { do_transition = true; break; }
    std::vector<int32_t> cat;
    std::vector<size_t> seg_end;
    std::vector<uint32_t> seg_lod, seg_map, seg_off;
    auto push = [&](size_t lod, size_t tile, size_t mi) {
        const auto& rd = w.tb(lod, tile, view_id).raw_depth;
        cat.insert(cat.end(), rd.begin(), rd.end());
        seg_end.push_back(cat.size());
        seg_lod.push_back((uint32_t)lod); seg_map.push_back((uint32_t)mi); seg_off.push_back(w.merge_offset[lod][tile]);
    };
    for (size_t mi : from_vec) {
        const TileInstance* mt = w.tile_map[mi].get();
        push(mt->lod, mt->tile, mi);
        if (mt->transition == TR_CHANGING_LOWER) push(mt->lod + 1, mt->tile, mi);
        else if (mt->transition == TR_CHANGING_HIGHER) push(mt->lod - 1, mt->tile, mi);
    }
    std::vector<uint32_t> order(cat.size());
    sort_raw_depth(cat.data(), cat.size(), order.data());
    val.splat_count = cat.size();
    val.gs_index.resize(cat.size()); val.gs_map_id.resize(cat.size());
    val.gs_lod_id.clear();
    if (do_transition) val.gs_lod_id.resize(cat.size());
    for (size_t j = 0; j < order.size(); j++) {
        size_t o = order[j];
        size_t seg = (size_t)(std::upper_bound(seg_end.begin(), seg_end.end(), o) - seg_end.begin());
        size_t start = seg == 0 ? 0 : seg_end[seg - 1];
        val.gs_index[j] = (uint32_t)(o - start) + seg_off[seg];
        val.gs_map_id[j] = seg_map[seg];
        if (do_transition) val.gs_lod_id[j] = seg_lod[seg];
    }
    val.merge_from_vec = from_vec;
    val.single_lod_id = do_transition ? -1 : (int32_t)head_lod;
    val.has_lod = do_transition;
}

std::string cache_key(size_t view_id, const std::vector<std::pair<size_t, size_t>>& tid, const std::vector<int>& st)
{
    std::string k;
    auto put = [&](uint64_t v) { k.append((const char*)&v, 8); };
    put(view_id); put(tid.size());
    for (auto& t : tid) { put(t.first); put(t.second); }
    for (int s : st) put((uint64_t)status_hash(s));
    return k;
}

int check_user(const gswt_user_data& u)
{
    if (u.surface_type == SURFACE_SPHERE && (u.tile_map_half_wh[0] == 0 || u.tile_map_half_wh[0] * 2 * 2 != u.tile_map_half_wh[1] * 2 * 5))
        return fail(GSWT_ERR_BAD_ARG, "WangTile::configure: assertion failed: tile_map_wh.x * 2 == tile_map_wh.y * 5 (Sphere maps are 5 : 2)");
    if (u.surface_type > 2 || u.tile_sort_type > 3 || u.merge_type > 2 || u.height_map_type > 4) return fail(GSWT_ERR_BAD_ARG, "WangTile::configure: enum out of range");
    if (!(u.tile_width > 0.0f)) return fail(GSWT_ERR_BAD_ARG, "WangTile::configure: tile_width must be positive");
    if (u.center_option == 0) return fail(GSWT_ERR_BAD_ARG, "WangTile::configure: center_option must be >= 1");
    if (u.cache_size == 0) return fail(GSWT_ERR_BAD_ARG, "WangTile::configure: cache_size must be nonzero (NonZeroUsize::new(..).unwrap())");
    return GSWT_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* gswt_host_last_error(void) { return g_err.c_str(); }

uint32_t gswt_pack_half_2x16(float x, float y) { return pack_half_2x16(x, y); }

int gswt_tileset_create(int n_lod, int n_tile, gswt_tileset** out)
try {
    if (!out || n_lod <= 0 || n_tile <= 0) return fail(GSWT_ERR_BAD_ARG, "gswt_tileset_create: bad dims");
    auto* ts = new gswt_tileset();
    ts->n_lod = n_lod; ts->n_tile = n_tile;
    ts->s.assign(n_lod, std::vector<Scene>(n_tile));
    *out = ts;
    return GSWT_OK;
} GSWT_CATCH("gswt_tileset_create")

void gswt_tileset_destroy(gswt_tileset* ts) { delete ts; }

static int ts_slot(gswt_tileset* ts, int lod, int tile, Scene** sc)
{
    if (!ts || lod < 0 || lod >= ts->n_lod || tile < 0 || tile >= ts->n_tile) return fail(GSWT_ERR_BAD_ARG, "tile set index out of range");
    *sc = &ts->s[lod][tile];
    return GSWT_OK;
}

int gswt_tileset_set_vertices(gswt_tileset* ts, int lod, int tile, const float* verts62, size_t n)
try {
    Scene* sc = nullptr;
    int rc = ts_slot(ts, lod, tile, &sc);
    if (rc) return rc;
    if (!verts62 && n) return fail(GSWT_ERR_BAD_ARG, "null vertices");
    scene_load(*sc, verts62, n);
    return GSWT_OK;
} GSWT_CATCH("gswt_tileset_set_vertices")

int gswt_tileset_set_ply(gswt_tileset* ts, int lod, int tile, const uint8_t* bytes, size_t len)
try {
    Scene* sc = nullptr;
    int rc = ts_slot(ts, lod, tile, &sc);
    if (rc) return rc;
    if (!bytes) return fail(GSWT_ERR_BAD_ARG, "null ply");
    size_t hs = 0, n = 0;
    rc = parse_ply_header(bytes, len, &hs, &n);
    if (rc) return rc;
    if (hs > 65535) return fail(GSWT_ERR_IO, "Scene::parse_file_header(): header of %zu bytes overflows the reference's u16", hs);
    if (hs > len || n > (len - hs) / 248) return fail(GSWT_ERR_IO, "Scene::load(): file holds fewer than %zu vertices (read_exact fails)", n);
    std::vector<float> verts(62 * n);
    if (n) memcpy(verts.data(), bytes + hs, n * 248);     // (an empty vector's data() may be null: memcpy must not see it even for 0 bytes)
    scene_load(*sc, verts.data(), n);
    return GSWT_OK;
} GSWT_CATCH("gswt_tileset_set_ply")

int gswt_tileset_set_rows(gswt_tileset* ts, int lod, int tile, const uint8_t* rows32, size_t n)
try {
    Scene* sc = nullptr;
    int rc = ts_slot(ts, lod, tile, &sc);
    if (rc) return rc;
    if (!rows32 && n) return fail(GSWT_ERR_BAD_ARG, "null rows");
    if (n > ((size_t)1 << 28)) return fail(GSWT_ERR_CAPACITY, "gswt_tileset_set_rows: %zu splats exceed 2^28", n);
    sc->splat_count = n;
    sc->buffer.assign(rows32, rows32 + 32 * n);
    return GSWT_OK;
} GSWT_CATCH("gswt_tileset_set_rows")

int gswt_load_scene_zip_mem(const uint8_t* bytes, size_t len, gswt_tileset** out)
try {
    if (!bytes || !out) return fail(GSWT_ERR_BAD_ARG, "gswt_load_scene_zip: null argument");
    std::vector<ZipEntry> all;
    int rc = zip_list(bytes, len, all);
    if (rc) return rc;
    struct F { size_t index; std::string filename; size_t lod, tile; };
    std::vector<F> files;
    std::regex re("lod(\\d+)_tile_(\\d+)");
    for (size_t i = 0; i < all.size(); i++) {
        std::string fname = all[i].name;
        size_t slash = fname.find_last_of('/');
        if (slash != std::string::npos) fname = fname.substr(slash + 1);
        if (fname.empty()) continue;
        std::smatch m;
        if (std::regex_search(fname, m, re)) {
            // the reference parses into usize and would then index / allocate by it; a tile set is a few LODs x 16..4096 tiles
            if (m[1].length() > 4 || m[2].length() > 7) return fail(GSWT_ERR_IO, "load_scene_zip: lod / tile number out of range in '%s'", fname.c_str());
            files.push_back({i, fname, (size_t)std::stoull(m[1]), (size_t)std::stoull(m[2])});
        }
    }
    if (files.empty()) return fail(GSWT_ERR_IO, "load_scene_zip: no lod<L>_tile_<T> entries");
    std::stable_sort(files.begin(), files.end(), [](const F& a, const F& b) { return a.lod != b.lod ? a.lod < b.lod : a.tile < b.tile; });
    size_t n_lod = files.back().lod - files.front().lod + 1, n_tile = files.back().tile + 1;
    if (n_lod > 64 || n_tile > ((size_t)1 << 20)) return fail(GSWT_ERR_IO, "load_scene_zip: %zu LODs x %zu tiles is not a tile set", n_lod, n_tile);
    if (files.size() < n_lod * n_tile) return fail(GSWT_ERR_IO, "load_scene_zip: %zu entries for %zu x %zu tiles (index out of bounds in the reference)", files.size(), n_lod, n_tile);
    gswt_tileset* ts = nullptr;
    rc = gswt_tileset_create((int)n_lod, (int)n_tile, &ts);
    if (rc) return rc;
    std::vector<uint8_t> data;
    for (size_t i = 0; i < n_lod; i++)
        for (size_t j = 0; j < n_tile; j++) {
            const F& f = files[i * n_tile + j];
            if (f.filename.find(".ply") != std::string::npos) {
                rc = zip_read(bytes, len, all[f.index], data);
                if (!rc) rc = gswt_tileset_set_ply(ts, (int)i, (int)j, data.data(), data.size());
                if (rc) { gswt_tileset_destroy(ts); return rc; }
            } else if (f.filename.find(".splat") != std::string::npos) {
                // scene.rs:1120-1125: the bytes are read but never stored -> empty scene
            } else { gswt_tileset_destroy(ts); return fail(GSWT_ERR_IO, "load_scene_zip: unreachable!() for %s", f.filename.c_str()); }
        }
    *out = ts;
    return GSWT_OK;
} GSWT_CATCH("gswt_load_scene_zip_mem")

int gswt_load_scene_zip(const char* path, gswt_tileset** out)
try {
    if (!path) return fail(GSWT_ERR_BAD_ARG, "null path");
    std::ifstream fs(path, std::ios::binary);
    if (!fs) return fail(GSWT_ERR_IO, "cannot open %s", path);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(fs)), std::istreambuf_iterator<char>());
    return gswt_load_scene_zip_mem(bytes.data(), bytes.size(), out);
} GSWT_CATCH("gswt_load_scene_zip")

int gswt_tileset_dims(const gswt_tileset* ts, int* n_lod, int* n_tile)
try {
    if (!ts) return GSWT_ERR_BAD_ARG;
    if (n_lod) *n_lod = ts->n_lod;
    if (n_tile) *n_tile = ts->n_tile;
    return GSWT_OK;
} GSWT_CATCH("gswt_tileset_dims")
size_t gswt_tileset_splat_count(const gswt_tileset* ts, int lod, int tile) { return ts->s[lod][tile].splat_count; }
const uint8_t* gswt_tileset_rows(const gswt_tileset* ts, int lod, int tile) { return ts->s[lod][tile].buffer.data(); }

int gswt_generate_texture(const uint8_t* rows32, size_t n, uint32_t* tex_out)
try {
    if ((!rows32 || !tex_out) && n) return GSWT_ERR_BAD_ARG;
    generate_texture(rows32, n, tex_out);
    return GSWT_OK;
} GSWT_CATCH("gswt_generate_texture")

int gswt_sort_raw_depth(const int32_t* depths, size_t n, uint32_t* order_out)
try {
    if ((!depths || !order_out) && n) return GSWT_ERR_BAD_ARG;
    sort_raw_depth(depths, n, order_out);
    return GSWT_OK;
} GSWT_CATCH("gswt_sort_raw_depth")

int gswt_camera_uniforms_from_camera(const float pos[3], const float target[3], const float up[3], float fovy_deg, float z_near,
                         float z_far, int width, int height, gswt_camera_uniforms* out, float* view_proj16)
try {
    if (!pos || !target || !up || !out || width <= 0 || height <= 0) return fail(GSWT_ERR_BAD_ARG, "gswt_camera_uniforms_from_camera: bad argument");
    float view[16], proj[16];
    look_at_rh(V3{pos[0], pos[1], pos[2]}, V3{target[0], target[1], target[2]}, V3{up[0], up[1], up[2]}, view);
    perspective(fovy_deg, (float)width / (float)height, z_near, z_far, proj);
    float w = (float)width, h = (float)height;
    float fx = 0.5f * proj[0] * w, fy = -0.5f * proj[5] * h;
    float fovy = fovy_deg * (float)(3.14159265358979323846 / 180.0);
    float htany = (float)std::tan((double)(fovy / 2.0f));
    float htanx = (htany / h) * w;
    memcpy(out->projection, proj, 64);
    memcpy(out->view, view, 64);
    out->focal[0] = std::fabs(fx); out->focal[1] = std::fabs(fy);
    out->viewport[0] = w; out->viewport[1] = h;
    out->htan_fov[0] = htanx; out->htan_fov[1] = htany; out->htan_fov[2] = 0; out->htan_fov[3] = 0;
    out->cam_pos[0] = pos[0]; out->cam_pos[1] = pos[1]; out->cam_pos[2] = pos[2]; out->cam_pos[3] = 0;
    if (view_proj16) mat4_mul(proj, view, view_proj16);
    return GSWT_OK;
} GSWT_CATCH("gswt_camera_uniforms_from_camera")

int gswt_wang_new(gswt_tileset* ts, gswt_wang** out)
try {
    if (!ts || !out) return fail(GSWT_ERR_BAD_ARG, "gswt_wang_new: null argument");
    auto* w = new gswt_wang();
    w->tiles = std::move(ts->s);
    delete ts;
    int rc = preprocess(*w);
    if (rc) { delete w; return rc; }
    *out = w;
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_new")

void gswt_wang_destroy(gswt_wang* w) { delete w; }

int gswt_wang_preload(gswt_wang* w, gswt_preload* out)
try {
    if (!w || !out) return GSWT_ERR_BAD_ARG;
    out->tex_data = w->tex.data();
    out->n_splats = w->merged_count;
    out->n_lod = (int)w->n_lod; out->n_tile = (int)w->n_tile; out->n_view = (int)w->n_view;
    out->lists = w->base_lists.data();
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_preload")

int gswt_wang_tile_base(const gswt_wang* w, int tile, float center[3], float aabb[6])
try {
    if (!w || tile < 0 || (size_t)tile >= w->n_tile) return GSWT_ERR_BAD_ARG;
    V3 c = w->tile_center[tile], lo = w->aabb_lo[tile], hi = w->aabb_hi[tile];
    if (center) { center[0] = c.x; center[1] = c.y; center[2] = c.z; }
    if (aabb) { aabb[0] = lo.x; aabb[1] = lo.y; aabb[2] = lo.z; aabb[3] = hi.x; aabb[4] = hi.y; aabb[5] = hi.z; }
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_tile_base")

int gswt_wang_lod_avg_scale(const gswt_wang* w, float* out, int cap)
try {
    if (!w || !out || cap < (int)w->n_lod) return GSWT_ERR_BAD_ARG;
    for (size_t l = 0; l < w->n_lod; l++) out[l] = w->lod_avg_scale[l];
    return (int)w->n_lod;
} GSWT_CATCH("gswt_wang_lod_avg_scale")

const int32_t* gswt_wang_raw_depth(const gswt_wang* w, int lod, int tile, int view, size_t* n)
{
    const auto& rd = w->tb(lod, tile, view).raw_depth;
    if (n) *n = rd.size();
    return rd.data();
}

int gswt_wang_merge_offset(const gswt_wang* w, int lod, int tile, uint32_t* out)
try {
    if (!w || !out || lod < 0 || (size_t)lod >= w->n_lod || tile < 0 || (size_t)tile >= w->n_tile) return GSWT_ERR_BAD_ARG;
    *out = w->merge_offset[lod][tile];
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_merge_offset")

int gswt_wang_configure(gswt_wang* w, const gswt_user_data* user, gswt_configured* out)
try {
    if (!w || !user) return fail(GSWT_ERR_BAD_ARG, "gswt_wang_configure: null argument");
    int rc = check_user(*user);
    if (rc) return rc;
    if (w->n_tile / 16 < user->center_option) return fail(GSWT_ERR_BAD_ARG, "WangTile::configure: assertion failed: self.n_tiles.1 / 16 >= self.user_data.center_option");
    w->initialized = false;
    w->user = *user;
    if (user->height_tex && user->height_tex_wh[0] && user->height_tex_wh[1]) {
        w->height_tex_copy.assign(user->height_tex, user->height_tex + (size_t)user->height_tex_wh[0] * user->height_tex_wh[1]);
        w->user.height_tex = w->height_tex_copy.data();
    } else { w->height_tex_copy.clear(); w->user.height_tex = nullptr; }
    if (w->user.reset_rng) w->rng.reseed(0);
    const int odd = w->user.surface_type == SURFACE_SPHERE ? 0 : 1;          // wangtile.rs:356-361
    w->map_w = (int)w->user.tile_map_half_wh[0] * 2 + odd;
    w->map_h = (int)w->user.tile_map_half_wh[1] * 2 + odd;
    w->tile_map.clear();
    w->tile_map.resize((size_t)w->map_w * w->map_h);
    w->neighbor_map.assign((size_t)w->map_w * w->map_h * 4, Neighbor{});
    for (int i = 0; i < w->map_w; i++)
        for (int j = 0; j < w->map_h; j++) compute_map_neighbors(*w, i, j, &w->neighbor_map[((size_t)i * w->map_h + j) * 4]);
    // height map, :376-413
    int hw = (int)w->user.height_map_wh[0], hh = (int)w->user.height_map_wh[1];
    std::vector<float> hm;
    for (int i = 0; i < hh; i++)
        for (int j = 0; j < hw; j++) {
            float h = 0.0f;
            switch (w->user.height_map_type) {
            case HMAP_TEXTURE: h = 0.0f; break;
            case HMAP_RANDOM: h = w->rng.random_range_f32_inclusive(-1.0f, 1.0f); break;
            case HMAP_SLOPEX: h = (float)j / (float)hh * 2.0f - 1.0f; break;
            case HMAP_SLOPEY: h = (float)i / (float)hh * 2.0f - 1.0f; break;
            default: h = ((float)i / (float)hw + (float)j / (float)hh) - 1.0f; break;
            }
            hm.push_back(h);
        }
    if (w->user.height_map_type == HMAP_TEXTURE && w->user.height_tex) {
        hw = (int)w->user.height_tex_wh[0]; hh = (int)w->user.height_tex_wh[1];
        hm = w->height_tex_copy;
    }
    {
        float k = w->user.tile_width * w->user.height_map_scale[2];
        for (float& v : hm) v *= k;
    }
    if (w->user.height_map_type == HMAP_RANDOM) {
        if (hw <= 0 || hh <= 0) return fail(GSWT_ERR_BAD_ARG, "WangTile::configure: Random height map needs height_map_wh > 0");
        hm = map_resize(hm, hw, hh, 1024, 1024);
        hw = hh = 1024;
    }
    if (w->user.surface_type == SURFACE_HEIGHTMAP && (hw <= 0 || hh <= 0)) return fail(GSWT_ERR_BAD_ARG, "WangTile::configure: HeightMap surface without a height map");
    w->height_map = std::move(hm);
    w->hm_w = hw; w->hm_h = hh;
    w->user.height_map_wh[0] = (uint32_t)hw; w->user.height_map_wh[1] = (uint32_t)hh;
    // lod transition dist, :415-423
    w->lod_transition_dist.clear();
    float s_n = w->lod_avg_scale.back();
    for (float s : w->lod_avg_scale) w->lod_transition_dist.push_back(w->user.lod_max_dist * s / s_n);
    w->lru.clear(); w->lru_index.clear();
    w->center_x = w->center_y = 0;     // (the reference keeps the previous center_coord; the map is empty so nothing is reused)
    if (out) {
        memset(out, 0, sizeof(*out));
        out->tile_map_wh[0] = (uint32_t)w->map_w; out->tile_map_wh[1] = (uint32_t)w->map_h;
        out->height_map_wh[0] = (uint32_t)hw; out->height_map_wh[1] = (uint32_t)hh;
        out->height_map = w->height_map.empty() ? nullptr : w->height_map.data();
        for (size_t l = 0; l < w->lod_transition_dist.size() && l < 16; l++) out->lod_transition_dist[l] = w->lod_transition_dist[l];
        out->n_lod = (uint32_t)w->n_lod; out->n_tile = (uint32_t)w->n_tile; out->n_view = (uint32_t)w->n_view;
    }
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_configure")

int gswt_wang_check_update(const gswt_wang* w, const float cam_pos[3])
try {
    if (!w || !cam_pos) return GSWT_ERR_BAD_ARG;
    if (!w->initialized) return 1;
    return distance2(V3{cam_pos[0], cam_pos[1], cam_pos[2]}, w->camera_pos) >= w->user.update_distance2 ? 1 : 0;
} GSWT_CATCH("gswt_wang_check_update")

int gswt_wang_build_tiles(gswt_wang* w, const float cam_pos[3], gswt_scene_data* out)
try {
    if (!w || !cam_pos) return fail(GSWT_ERR_BAD_ARG, "gswt_wang_build_tiles: null argument");
    if (w->map_w == 0) return fail(GSWT_ERR_STATE, "WangTile::build_tiles before configure");
    w->initialized = true;
    update_tile_map(*w, V3{cam_pos[0], cam_pos[1], cam_pos[2]});
    if (out) {
        memset(out, 0, sizeof(*out));
        out->center_coord[0] = w->center_x; out->center_coord[1] = w->center_y;
        for (int i = 0; i < w->map_w; i++)
            for (int j = 0; j < w->map_h; j++) {
                const TileInstance* ti = w->at(i, j);
                size_t c = w->tb(ti->lod, ti->tile, 0).splat_count;
                out->splat_count += c; out->blending_splat_count += c;
                out->lod_splat_count[ti->lod] += c; out->lod_instance_count[ti->lod] += 1;
                bool blend_lower = ti->lod < w->n_lod - 1;
                if (ti->transition == TR_CHANGING_HIGHER) { out->blending_splat_count += w->tb(ti->lod - 1, ti->tile, 0).splat_count; blend_lower = false; }
                if (blend_lower) out->blending_splat_count += w->tb(ti->lod + 1, ti->tile, 0).splat_count;
            }
    }
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_build_tiles")

int gswt_wang_set_device_merge(gswt_wang* w, int enable)
try {
    if (!w) return GSWT_ERR_BAD_ARG;
    w->device_merge = enable != 0;
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_set_device_merge")

int gswt_wang_raw_depth_tables(gswt_wang* w, const int32_t* const** ptrs, const uint32_t** counts, const uint32_t** merge_offset)
try {
    if (!w || !ptrs || !counts || !merge_offset) return GSWT_ERR_BAD_ARG;
    w->rd_ptrs.clear(); w->rd_counts.clear(); w->rd_offsets.clear();
    for (size_t l = 0; l < w->n_lod; l++)
        for (size_t t = 0; t < w->n_tile; t++) {
            w->rd_counts.push_back((uint32_t)w->tiles[l][t].splat_count);
            w->rd_offsets.push_back(w->merge_offset[l][t]);
            for (size_t v = 0; v < w->n_view; v++) w->rd_ptrs.push_back(w->tb(l, t, v).raw_depth.data());
        }
    *ptrs = w->rd_ptrs.data(); *counts = w->rd_counts.data(); *merge_offset = w->rd_offsets.data();
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_raw_depth_tables")

int gswt_wang_worker_config(gswt_wang* w, gswt_worker_config* out)
try {
    if (!w || !out) return fail(GSWT_ERR_BAD_ARG, "gswt_wang_worker_config: null argument");
    if (w->map_w == 0) return fail(GSWT_ERR_STATE, "gswt_wang_worker_config before configure");
    memset(out, 0, sizeof(*out));
    const gswt_user_data& u = w->user;
    out->map_w = (uint32_t)w->map_w; out->map_h = (uint32_t)w->map_h;
    out->half_w = u.tile_map_half_wh[0]; out->half_h = u.tile_map_half_wh[1];
    out->n_lod = (uint32_t)w->n_lod; out->n_tile = (uint32_t)w->n_tile; out->n_view = (uint32_t)w->presort_dirs.size();
    out->tile_width = u.tile_width;
    out->surface_type = u.surface_type; out->tile_sort_type = u.tile_sort_type; out->merge_type = u.merge_type;
    for (int k = 0; k < 3; k++) out->height_map_scale[k] = u.height_map_scale[k];
    out->sphere_radius = u.sphere_radius;
    out->lod_blending = u.lod_blending; out->lod_bbox_check = u.lod_bbox_check;
    out->lod_transition_width_ratio = u.lod_transition_width_ratio; out->lod_dist_tolerance = u.lod_dist_tolerance;
    out->merge_tile_dist[0] = u.merge_tile_dist[0]; out->merge_tile_dist[1] = u.merge_tile_dist[1];
    out->merge_dot_threshold = u.merge_dot_threshold; out->merge_topk = u.merge_topk;
    out->hm_w = (uint32_t)w->hm_w; out->hm_h = (uint32_t)w->hm_h;
    out->height_map = w->height_map.empty() ? nullptr : w->height_map.data();
    out->lod_transition_dist = w->lod_transition_dist.data();
    w->wk_center.clear(); w->wk_aabb.clear(); w->wk_dirs.clear(); w->wk_counts.clear(); w->wk_nb.clear();
    for (size_t t = 0; t < w->n_tile; t++) {
        const V3 c = w->tile_center[t], lo = w->aabb_lo[t], hi = w->aabb_hi[t];
        w->wk_center.insert(w->wk_center.end(), {c.x, c.y, c.z});
        w->wk_aabb.insert(w->wk_aabb.end(), {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z});
    }
    for (size_t l = 0; l < w->n_lod; l++)
        for (size_t t = 0; t < w->n_tile; t++) w->wk_counts.push_back((uint32_t)w->tb(l, t, 0).raw_depth.size());
    for (const V3& d : w->presort_dirs) w->wk_dirs.insert(w->wk_dirs.end(), {d.x, d.y, d.z});
    for (int x = 0; x < w->map_w; x++)
        for (int y = 0; y < w->map_h; y++)
            for (int s = 0; s < 4; s++) {
                const Neighbor& n = w->nb(x, y, s);
                w->wk_nb.push_back(n.some ? (int32_t)((w->map_to_index(n.x, n.y) << 2) | (size_t)n.slot) : -1);
            }
    out->tile_center = w->wk_center.data(); out->tile_aabb = w->wk_aabb.data(); out->splat_count = w->wk_counts.data();
    out->presort_dirs = w->wk_dirs.data(); out->neighbors = w->wk_nb.data();
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_worker_config")

int gswt_wang_export_cells(const gswt_wang* w, gswt_cell* out, size_t cap, int32_t center_coord[2])
try {
    if (!w || !out) return fail(GSWT_ERR_BAD_ARG, "gswt_wang_export_cells: null argument");
    const size_t n = (size_t)w->map_w * w->map_h;
    if (cap < n) return fail(GSWT_ERR_CAPACITY, "need %zu cells", n);
    for (size_t i = 0; i < n; i++) {
        const TileInstance* t = w->tile_map[i].get();
        if (!t) return fail(GSWT_ERR_STATE, "tile map not built");
        gswt_cell& c = out[i];
        memset(&c, 0, sizeof(c));
        c.tile = (uint32_t)t->tile; c.has_corner = t->has_corner ? 1u : 0u;
        c.tile_offset[0] = t->tile_offset.x; c.tile_offset[1] = t->tile_offset.y; c.tile_offset[2] = t->tile_offset.z;
        c.tile_center[0] = t->tile_center.x; c.tile_center[1] = t->tile_center.y; c.tile_center[2] = t->tile_center.z;
        memcpy(c.to_local, t->to_local.m, sizeof(c.to_local));
        if (t->has_corner)
            for (int k = 0; k < 4; k++) {
                const V3 up = t->corner[k].to_world.col(2);
                c.corner_pos[3 * k] = t->corner[k].pos.x; c.corner_pos[3 * k + 1] = t->corner[k].pos.y; c.corner_pos[3 * k + 2] = t->corner[k].pos.z;
                c.corner_up[3 * k] = up.x; c.corner_up[3 * k + 1] = up.y; c.corner_up[3 * k + 2] = up.z;
                c.edge_pos[3 * k] = t->edge[k].pos.x; c.edge_pos[3 * k + 1] = t->edge[k].pos.y; c.edge_pos[3 * k + 2] = t->edge[k].pos.z;
                c.edge_normal[3 * k] = t->edge[k].normal.x; c.edge_normal[3 * k + 1] = t->edge[k].normal.y; c.edge_normal[3 * k + 2] = t->edge[k].normal.z;
            }
    }
    if (center_coord) { center_coord[0] = w->center_x; center_coord[1] = w->center_y; }
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_export_cells")

int gswt_wang_export_cell_state(const gswt_wang* w, gswt_cell_state* out, size_t cap)
try {
    if (!w || !out) return fail(GSWT_ERR_BAD_ARG, "gswt_wang_export_cell_state: null argument");
    const size_t n = (size_t)w->map_w * w->map_h;
    if (cap < n) return fail(GSWT_ERR_CAPACITY, "need %zu cells", n);
    for (size_t i = 0; i < n; i++) {
        const TileInstance* t = w->tile_map[i].get();
        if (!t) return fail(GSWT_ERR_STATE, "tile map not built");
        out[i].lod = (uint32_t)t->lod; out[i].transition = t->transition; out[i].spawning_factor = t->spawning;
        out[i].merge = (uint32_t)t->merge; out[i].merged_to = t->merge == MS_TO ? (uint32_t)t->merged_to : 0u;
    }
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_export_cell_state")

int gswt_wang_get_tile_ids(const gswt_wang* w, uint32_t* ids, size_t cap)
try {
    if (!w || !ids) return GSWT_ERR_BAD_ARG;
    size_t n = (size_t)w->map_w * w->map_h;
    if (cap < n) return fail(GSWT_ERR_CAPACITY, "need %zu ids", n);
    for (size_t i = 0; i < n; i++) {
        if (!w->tile_map[i]) return fail(GSWT_ERR_STATE, "tile map not built");
        ids[i] = (uint32_t)w->tile_map[i]->tile;
    }
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_get_tile_ids")

int gswt_wang_set_tile_ids(gswt_wang* w, const uint32_t* ids, size_t n)
try {
    if (!w || !ids) return GSWT_ERR_BAD_ARG;
    if (n != (size_t)w->map_w * w->map_h) return fail(GSWT_ERR_BAD_ARG, "expected %d ids", w->map_w * w->map_h);
    for (size_t i = 0; i < n; i++) {
        if (!w->tile_map[i]) return fail(GSWT_ERR_STATE, "tile map not built");
        if (ids[i] >= w->n_tile) return fail(GSWT_ERR_BAD_ARG, "tile id %u out of range", ids[i]);
    }
    // rebuild every instance with the given id, as update_tile_map would for a freshly spawned tile
    for (auto& t : w->tile_map) t->has_corner = false;
    for (int i = 0; i < w->map_w; i++)
        for (int j = 0; j < w->map_h; j++) {
            TileInstance* ti = w->at(i, j);
            ti->tile = ids[w->map_to_index(i, j)];
            V3 base_center = w->tile_center[ti->tile];
            surface_mapping(*w, i, j, base_center + ti->tile_offset, false, ti->tile_center, ti->to_local);
            ti->has_corner = compute_corner_edge(*w, i, j, base_center.z, ti->corner, ti->edge);
        }
    update_lod(*w, w->camera_pos);
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_set_tile_ids")

int gswt_wang_sort_tiles(gswt_wang* w, const float cam_pos[3], const float vp[16], gswt_sort_data* out)
try {
    if (!w || !cam_pos || !vp || !out) return fail(GSWT_ERR_BAD_ARG, "gswt_wang_sort_tiles: null argument");
    if (!w->initialized) return fail(GSWT_ERR_STATE, "WangTile::sort_tiles before build_tiles");
    const bool need_corner = w->user.tile_sort_type == SORT_GRAPH || w->user.merge_type == MERGE_EDGE;
    (void)need_corner;
    V3 cam{cam_pos[0], cam_pos[1], cam_pos[2]};
    if (w->user.merge_type == MERGE_AXIS) selective_merge_axis(*w, cam, vp);
    else if (w->user.merge_type == MERGE_EDGE) selective_merge_edge(*w, cam, vp);
    std::vector<size_t> order;
    switch (w->user.tile_sort_type) {
    case SORT_DISTANCE: order = sort_tiles_object_pos(*w, cam); break;
    case SORT_VIEWPORT: order = sort_tiles_object_vp(*w, vp); break;
    case SORT_OBJECT: order = sort_tiles_object_bfs(*w, cam); break;
    default: order = sort_tiles_object_graph(*w, cam); break;
    }
    w->sorted.clear();
    w->m_gs.clear(); w->m_map.clear(); w->m_lod.clear();
    w->m_groups.clear(); w->m_members.clear();
    size_t merged_total = 0;
    for (size_t mi : order) {
        const TileInstance* ti = w->tile_map[mi].get();
        size_t view_id;
        std::vector<std::pair<size_t, size_t>> tids;
        std::vector<int> sts;
        if (ti->merge == MS_FROM) {
            bool merge_x = true, merge_y = true;
            V3 avg_c;
            Quat avg_q;
            for (size_t m_mi : ti->merged_from) {
                int mx, my;
                w->index_to_map(m_mi, mx, my);
                if (mx != ti->mx) merge_x = false;
                if (my != ti->my) merge_y = false;
                const TileInstance* mt = w->tile_map[m_mi].get();
                tids.emplace_back(mt->lod, mt->tile);
                sts.push_back(mt->transition);
                avg_c = avg_c + mt->tile_center;
                Quat q = quat_from_mat3(mt->to_local);
                avg_q.s += q.s; avg_q.x += q.x; avg_q.y += q.y; avg_q.z += q.z;
            }
            if (!merge_x && !merge_y) view_id = w->presort_dirs.size() - 1;
            else {
                float n = (float)ti->merged_from.size();
                Quat q{avg_q.s / n, avg_q.x / n, avg_q.y / n, avg_q.z / n};
                view_id = choose_presort_view(*w, mat3_from_quat(q), avg_c / n, cam);
            }
        } else {
            view_id = choose_presort_view(*w, ti->to_local, ti->tile_center, cam);
            tids.emplace_back(ti->lod, ti->tile);
            sts.push_back(ti->transition);
        }
        gswt_sorted_tile st;
        memset(&st, 0, sizeof(st));
        st.lod = (uint32_t)ti->lod; st.tile = (uint32_t)ti->tile; st.view_id = (uint32_t)view_id;
        st.tile_offset[0] = ti->tile_offset.x; st.tile_offset[1] = ti->tile_offset.y; st.tile_offset[2] = ti->tile_offset.z;
        st.map_index = (uint32_t)ti->map_index;
        st.map_coord[0] = (uint32_t)ti->mx; st.map_coord[1] = (uint32_t)ti->my;
        st.tile_center[0] = ti->tile_center.x; st.tile_center[1] = ti->tile_center.y; st.tile_center[2] = ti->tile_center.z;
        st.transition = ti->transition; st.spawning_factor = ti->spawning;
        st.has_corners = ti->has_corner ? 1 : 0;
        for (int ci = 0; ci < 4; ci++) { st.corners[3 * ci] = ti->corner[ci].pos.x; st.corners[3 * ci + 1] = ti->corner[ci].pos.y; st.corners[3 * ci + 2] = ti->corner[ci].pos.z; }
        st.key_len = (uint32_t)tids.size();
        st.single_lod_id = -1;
        if (ti->merge == MS_FROM) {
            // group description (what gswt_set_draws_merge_groups consumes)
            gswt_merge_group grp;
            grp.view_id = (uint32_t)view_id; grp.first_member = (uint32_t)w->m_members.size(); grp.n_members = (uint32_t)ti->merged_from.size(); grp._pad = 0;
            size_t group_len = 0;
            bool do_transition = false;
            for (size_t m_mi : ti->merged_from) {
                const TileInstance* mt = w->tile_map[m_mi].get();
                gswt_merge_member mm;
                mm.map_index = (uint32_t)m_mi; mm.lod = (uint32_t)mt->lod; mm.tile = (uint32_t)mt->tile; mm.other_lod = -1;
                if (mt->transition == TR_CHANGING_LOWER) mm.other_lod = (int32_t)mt->lod + 1;
                else if (mt->transition == TR_CHANGING_HIGHER) mm.other_lod = (int32_t)mt->lod - 1;
                if (mt->transition != TR_NONE) do_transition = true;
                group_len += w->tb(mt->lod, mt->tile, view_id).raw_depth.size();
                if (mm.other_lod >= 0) group_len += w->tb((size_t)mm.other_lod, mt->tile, view_id).raw_depth.size();
                w->m_members.push_back(mm);
            }
            st.merged = 1;
            st.merged_group = (uint32_t)w->m_groups.size();
            st.merged_offset = (uint32_t)merged_total;
            st.merged_count = (uint32_t)group_len;
            st.single_lod_id = do_transition ? -1 : (int32_t)ti->lod;
            w->m_groups.push_back(grp);
            merged_total += group_len;
            if (!w->device_merge) {
            std::string key = cache_key(view_id, tids, sts);
            const RenderDataValue* val = nullptr;
            RenderDataValue fresh;
            std::vector<uint32_t> remapped;
            if (w->user.use_cache) {
                auto it = w->lru_index.find(key);
                if (it != w->lru_index.end()) {
                    w->lru.splice(w->lru.begin(), w->lru, it->second);     // LruCache::get promotes
                    val = &it->second->second;
                    st.cache_hit = 1;
                    // update map index, :578-589: first matching member wins
                    remapped = val->gs_map_id;
                    const auto& old = val->merge_from_vec;
                    for (size_t i = 0; i < val->splat_count; i++)
                        for (size_t j = 0; j < old.size(); j++)
                            if (remapped[i] == (uint32_t)old[j]) { remapped[i] = (uint32_t)ti->merged_from[j]; break; }
                }
            }
            if (!val) {
                build_merged_value(*w, ti->merged_from, view_id, ti->lod, fresh);
                if (w->user.use_cache) {
                    w->lru.emplace_front(key, fresh);                       // LruCache::put
                    w->lru_index[key] = w->lru.begin();
                    while (w->lru.size() > w->user.cache_size) { w->lru_index.erase(w->lru.back().first); w->lru.pop_back(); }
                }
                val = &fresh;
                remapped = fresh.gs_map_id;
            }
            st.merged_count = (uint32_t)val->splat_count;
            st.single_lod_id = val->single_lod_id;
            w->m_gs.insert(w->m_gs.end(), val->gs_index.begin(), val->gs_index.end());
            w->m_map.insert(w->m_map.end(), remapped.begin(), remapped.end());
            if (val->has_lod) w->m_lod.insert(w->m_lod.end(), val->gs_lod_id.begin(), val->gs_lod_id.end());
            else w->m_lod.insert(w->m_lod.end(), val->splat_count, 0u);
            }
        }
        w->sorted.push_back(st);
    }
    out->scene_id = 0;
    out->n_tiles = (uint32_t)w->sorted.size();
    out->tiles = w->sorted.data();
    out->n_merged = merged_total;
    out->merged_gs_index = w->device_merge ? nullptr : w->m_gs.data();
    out->merged_map_id = w->device_merge ? nullptr : w->m_map.data();
    out->merged_lod_id = w->device_merge ? nullptr : w->m_lod.data();
    out->n_groups = (uint32_t)w->m_groups.size(); out->n_members = (uint32_t)w->m_members.size();
    out->groups = w->m_groups.data(); out->members = w->m_members.data();
    return GSWT_OK;
} GSWT_CATCH("gswt_wang_sort_tiles")

// renderer.rs:466-591 (host half) + TileUniforms::from_tile :691-725
int gswt_renderer_build_draws(const gswt_sort_data* sort, gswt_draw* draws_out)
try {
    if (!sort || (!draws_out && sort->n_tiles)) return fail(GSWT_ERR_BAD_ARG, "gswt_renderer_build_draws: null argument");
    for (uint32_t i = 0; i < sort->n_tiles; i++) {
        const gswt_sorted_tile& t = sort->tiles[i];
        gswt_draw& d = draws_out[i];
        memset(&d, 0, sizeof(d));
        gswt_tile_uniforms& u = d.tile;
        u.single_draw = 0; u.map_index = t.map_index; u.single_lod_id = -1; u.valid_lod_id = -1; u.changing = 0; u.changing_to_lower = -1;
        u.tile_id[0] = t.lod; u.tile_id[1] = t.tile; u.tile_id[2] = t.view_id; u.tile_id[3] = 0;
        u.offset[0] = t.tile_offset[0]; u.offset[1] = t.tile_offset[1]; u.offset[2] = t.tile_offset[2]; u.offset[3] = 0.0f;
        u.map_coord[0] = t.map_coord[0]; u.map_coord[1] = t.map_coord[1];
        d.lod = t.lod;
        if (t.merged) {
            u.single_draw = 1;
            u.single_lod_id = t.single_lod_id;
            u.changing = t.single_lod_id == -1 ? 1u : 0u;
            d.merged = 1; d.merged_offset = t.merged_offset; d.merged_count = t.merged_count; d.merged_group = t.merged_group;
            d.merged_has_lod = t.single_lod_id == -1 ? 1u : 0u;
        } else {
            d.base_lod = t.lod;
            if (t.transition == TR_CHANGING_LOWER) { u.changing = 1; u.changing_to_lower = 1; }
            else if (t.transition == TR_CHANGING_HIGHER) {
                u.changing = 1; u.changing_to_lower = 0;
                if (t.lod == 0) return fail(GSWT_ERR_BAD_ARG, "render: Changing(false) on lod 0 (index underflow in the reference)");
                d.base_lod = t.lod - 1;
            } else u.valid_lod_id = (int32_t)t.lod;
            d.base_tile = t.tile; d.base_view = t.view_id;
        }
        if (t.key_len == 1) {                          // viewport culling only for non-merged tiles, :472
            if (!t.has_corners) return fail(GSWT_ERR_STATE, "render: corner_data is None (called `Option::unwrap()` on a `None` value, renderer.rs:476)");
            d.cull_enable = 1;
            memcpy(d.corners, t.corners, sizeof(d.corners));
        }
    }
    return GSWT_OK;
} GSWT_CATCH("gswt_renderer_build_draws")

int gswt_scene_uniforms_from_data(const gswt_user_data* user, const gswt_configured* conf, const gswt_scene_data* scene, float splat_scale,
                        const float scene_scale[3], float height_map_scale_v, gswt_scene_uniforms* out)
try {
    if (!user || !conf || !scene || !out) return fail(GSWT_ERR_BAD_ARG, "gswt_scene_uniforms_from_data: null argument");
    memset(out, 0, sizeof(*out));
    out->splat_scale = splat_scale;
    out->tile_width = user->tile_width;
    out->surface_type = user->surface_type;
    out->sphere_radius = user->sphere_radius;
    out->transition_width_ratio = user->lod_transition_width_ratio;
    out->num_lod = conf->n_tile;          // renderer.rs:646: `user_data.n_tiles.1` (the TILE count, reference quirk)
    out->map_half_wh[0] = user->tile_map_half_wh[0]; out->map_half_wh[1] = user->tile_map_half_wh[1];
    out->center_coord[0] = scene->center_coord[0]; out->center_coord[1] = scene->center_coord[1];
    for (int i = 0; i < 16; i++) out->transition_dist_vec[i] = i < (int)conf->n_lod ? conf->lod_transition_dist[i] : 0.0f;
    out->height_map_scale[0] = user->height_map_scale[0];
    out->height_map_scale[1] = user->height_map_scale[1];
    out->height_map_scale[2] = user->height_map_scale[2] * height_map_scale_v;
    out->scene_scale[0] = scene_scale ? scene_scale[0] : 1.0f;
    out->scene_scale[1] = scene_scale ? scene_scale[1] : 1.0f;
    out->scene_scale[2] = scene_scale ? scene_scale[2] : 1.0f;
    return GSWT_OK;
} GSWT_CATCH("gswt_scene_uniforms_from_data")

}  // extern "C"
