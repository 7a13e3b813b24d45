#!/usr/bin/env python3
"""Insert wall_clock64 stamps into the two Lloyd pass kernels of csrc/kmeans.hip IN PLACE (debugging aid, never committed):
    python tools/dbg/apply_pass_stamps.py && bash tools/build_variant.sh stamps && git checkout gabor_color_image_segmentation_amd/csrc/kmeans.hip
    python tools/dbg/nv_stamps.py build_ab/stamps.so [n_scales n_orient]
Anchors are lines of the kernels; an anchor that no longer matches raises (update it with the kernel)."""
import os
p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                 "gabor_color_image_segmentation_amd", "csrc", "kmeans.hip")
s = open(p).read()
pre = '''__device__ unsigned long long g_nv_stamps[1024 * 16];
extern "C" int gcs_debug_nv_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nv_stamps), sizeof(unsigned long long) * 1024 * 16);
}
#define NV_STAMP(k) do { if (tid == 0) g_nv_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = wall_clock64(); } while (0)
#define NV_XCC() do { if (tid == 0) g_nv_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 15] = __builtin_amdgcn_s_getreg(20 | (3 << 11)); } while (0)
'''
k0 = s.index("template <typename T>\n")
s = s[:k0] + pre + s[k0:]
k = s.index("// MINB = workgroups per CU the register budget is set for")
head, tail = s[:k], s[k:]


def ins(txt, anchor, stamp, before=True):
    assert txt.count(anchor) == 1, (txt.count(anchor), anchor)
    return txt.replace(anchor, stamp + anchor if before else anchor + stamp)


# kmeans_pass_mfma_kernel
head = ins(head, "    uint16_t *cs = reinterpret_cast<uint16_t *>(s_tile);\n    for (int i = tid; i < 8 * KT * KP_ROWS; i += NTHR) {", "    NV_STAMP(0); NV_XCC();\n")
head = ins(head, "    // ---- per-cluster key base (exact int64)", "    NV_STAMP(3);\n")
head = ins(head, "    // ---- assign A fragments: row r = 4*jj + pat of tile mt", "    NV_STAMP(4);\n")
head = ins(head, "    __syncthreads();                                   // scratch reads done: the tile buffer is free again\n", "    NV_STAMP(5);\n")
head = ins(head, "    __syncthreads();                                   // scratch reads done: the tile buffer is free again\n", "    NV_STAMP(6);\n", before=False)
head = ins(head, "    if (!do_acc) return;\n    // ---- fold the four waves' accumulators", "    NV_STAMP(8);\n")
head = ins(head, "        partials[partial_index(per_image, b, part, parts, (int)gridDim.y, i, K * D1)] = (uint64_t)out;\n    }\n}\n\n// ----------------------------------------------------------------------------"
           "-----------\n// One Lloyd pass for DEEP banks", "", before=True)
head = head.replace("        partials[partial_index(per_image, b, part, parts, (int)gridDim.y, i, K * D1)] = (uint64_t)out;\n    }\n}\n\n// -----------------------------------------------------------"
                    "----------------------------\n// One Lloyd pass for DEEP banks",
                    "        partials[partial_index(per_image, b, part, parts, (int)gridDim.y, i, K * D1)] = (uint64_t)out;\n    }\n    __syncthreads();\n    NV_STAMP(9);\n}\n\n// ------------------"
                    "---------------------------------------------------------------------\n// One Lloyd pass for DEEP banks")
# kmeans_pass_native_kernel
tail = ins(tail, "    const bool working = part < parts_eff;", "    NV_STAMP(0); NV_XCC();\n")
tail = ins(tail, "    int ltile = g;\n    if (ltile < nlist) stage_load(phys(ltile));\n", "    NV_STAMP(1);\n", before=False)
tail = ins(tail, "    __syncthreads();\n    for (int j = tid >> 4; j < 16; j += 16) {                // key base", "    NV_STAMP(2);\n")
tail = ins(tail, "    for (int j = tid >> 4; j < 16; j += 16) {                // key base", "    NV_STAMP(3);\n")
tail = ins(tail, "    const int a_jj = (lane & 31) >> 2;", "    NV_STAMP(4);\n")
tail = ins(tail, "    __syncthreads();                                   // scratch reads done: the tile buffer is free\n", "    NV_STAMP(5);\n")
tail = ins(tail, "    __syncthreads();                                   // scratch reads done: the tile buffer is free\n", "    NV_STAMP(6);\n", before=False)
tail = ins(tail, "    if (!do_acc) return;\n\n    // ---- fold, every level at once", "    NV_STAMP(8);\n")
tail = ins(tail, "        partials[prow(i)] = (uint64_t)out;\n    }\n}\n\nstatic size_t assign_lds_bytes", "")
tail = tail.replace("        partials[prow(i)] = (uint64_t)out;\n    }\n}\n\nstatic size_t assign_lds_bytes",
                    "        partials[prow(i)] = (uint64_t)out;\n    }\n    __syncthreads();\n    NV_STAMP(9);\n}\n\nstatic size_t assign_lds_bytes")
open(p, "w").write(head + tail)
print("stamps inserted into", p)
