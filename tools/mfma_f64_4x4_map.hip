// Development aid: operand / result lane maps of v_mfma_f64_4x4x4_4b_f64, found by experiment (one-hot operands).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_4x4_map.hip -o mfma_map && ./mfma_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void probe(int *where)
{
    const int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    if (d != 0.0) where[blockIdx.x] = lane;
}

int main()
{
    int *d_where;
    std::vector<int> where(4096, -1);
    (void)hipMalloc(&d_where, 4096 * sizeof(int));
    (void)hipMemcpy(d_where, where.data(), 4096 * sizeof(int), hipMemcpyHostToDevice);
    probe<<<4096, 64>>>(d_where);
    (void)hipMemcpy(where.data(), d_where, 4096 * sizeof(int), hipMemcpyDeviceToHost);
    // hypothesis: A lane = i + 4 k + 16 b, B lane = j + 4 k + 16 b, D lane = j + 4 i + 16 b
    int bad = 0, hits = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const int ia = la % 4, ka = (la / 4) % 4, ba = la / 16, jb = lb % 4, kb = (lb / 4) % 4, bb = lb / 16;
            const int expect = (ka == kb && ba == bb) ? jb + 4 * ia + 16 * ba : -1;
            const int got = where[la * 64 + lb];
            hits += got >= 0;
            if (got != expect) {
                if (bad < 24) printf("A lane %2d x B lane %2d -> D lane %2d, hypothesis says %2d\n", la, lb, got, expect);
                ++bad;
            }
        }
    printf("%d non-zero products, %d disagree with: A lane = i + 4 k + 16 b, B lane = j + 4 k + 16 b, D lane = j + 4 i + 16 b\n",
           hits, bad);
    for (int la = 0; la < 64; ++la) {
        printf("A lane %d meets B lanes:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (where[la * 64 + lb] >= 0) printf(" %d->D%d", lb, where[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
