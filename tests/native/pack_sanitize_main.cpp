// Host packer under AddressSanitizer + UBSan (tests/test_pack_sanitize.py): every compiled network description is sized, mapped and
// packed from random weights; buffers one element short must be refused, not overrun.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include "nefes_hip.h"
int main() {
    int bad = 0;
    for (int w : {128, 256}) for (int c : {16, 128}) for (int tr : {0, 1}) for (int enc : {0, 1}) {
        NefesNetDesc d{w, c, tr, enc};
        NefesBlobInfo info;
        int rc = nefes_blob_info(&d, &info);
        if (rc) { printf("W=%d C=%d tr=%d enc=%d: unsupported (%d)\n", w, c, tr, enc, rc); continue; }
        int64_t elems[36];
        std::vector<uint32_t> map(info.total_bytes / 2);
        rc = nefes_pack_map(&d, map.data(), (int64_t)map.size(), elems);
        const int n = tr ? 36 : 24;
        std::vector<std::vector<float>> t(n);
        std::vector<const float*> ptr(n);
        for (int i = 0; i < n; ++i) { t[i].resize(elems[i]); for (auto& v : t[i]) v = (float)rand() / RAND_MAX - 0.5f; ptr[i] = t[i].data(); }
        std::vector<char> blob(info.total_bytes);
        int rc2 = nefes_pack_weights(&d, ptr.data(), n, blob.data(), blob.size());
        printf("W=%d C=%d tr=%d enc=%d: %llu bytes, map rc %d, pack rc %d\n", w, c, tr, enc, (unsigned long long)info.total_bytes, rc, rc2);
        bad += (rc != 0) + (rc2 != 0);
        // too-small buffers must be refused, not overrun
        if (nefes_pack_weights(&d, ptr.data(), n, blob.data(), blob.size() - 1) == 0) { printf("  short blob accepted\n"); ++bad; }
        if (nefes_pack_map(&d, map.data(), (int64_t)map.size() - 2, elems) == 0) { printf("  short map accepted\n"); ++bad; }
    }
    return bad;
}
