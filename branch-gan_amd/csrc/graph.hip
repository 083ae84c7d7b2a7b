// hipGraph hygiene for the captured train step (include/branchgan.h: bg_graph_node_census, bg_graph_replace_memset_nodes).
//
// Measured on MI355X, ROCm 7.2.0 (benchmarks/probes/graph_anomalies.py memset): a MEMSET node of a captured graph (what a
// captured hipMemsetAsync becomes) clears its destination on the first replay of the instantiated graph and fills it with
// garbage from the second replay on, for every size >= 16 bytes tried (16 B .. 4 KiB; 4 B is fine).  ATen's multi-block
// reductions clear their arrival counters ("semaphores") with exactly such a call before every launch, so from the second
// replay no workgroup finds itself the last one and the reduction's result is never written: the NaN / stale bias gradients
// of round 1.  The cure is to keep memset nodes out of a graph before it is instantiated: every memset node is replaced by a
// kernel node that runs graph_fill_kernel with the same destination, value, element size, width, height and pitch, wired to
// the same predecessors and successors.
#include <vector>

#include "common.h"

namespace bg {

// One thread per element; rows of `width` elements, `pitch` bytes apart.  elem_size 1, 2 or 4 (hipMemsetParams).
__global__ void graph_fill_kernel(unsigned char* dst, unsigned value, unsigned elem_size, size_t width, size_t height, size_t pitch) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = blockIdx.y;
  if (i >= width || row >= height) return;
  unsigned char* p = dst + row * pitch + i * elem_size;
  if (elem_size == 4) *(unsigned*)p = value;
  else if (elem_size == 2) *(unsigned short*)p = (unsigned short)value;
  else *p = (unsigned char)value;
}

#define BG_HIP(call, what)                                                            \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      set_error("%s: %s failed: %s", who, what, hipGetErrorString(e_));               \
      return BG_ERR_LAUNCH;                                                           \
    }                                                                                 \
  } while (0)

static int list_nodes(const char* who, hipGraph_t graph, std::vector<hipGraphNode_t>& nodes) {
  size_t n = 0;
  BG_HIP(hipGraphGetNodes(graph, nullptr, &n), "hipGraphGetNodes");
  nodes.resize(n);
  if (n) BG_HIP(hipGraphGetNodes(graph, nodes.data(), &n), "hipGraphGetNodes");
  nodes.resize(n);
  return BG_OK;
}

}  // namespace bg

using namespace bg;

extern "C" int bg_graph_node_census(void* graph, int* n_nodes, int* n_kernel, int* n_memset, int* n_memcpy) {
  const char* who = "bg_graph_node_census";
  BG_REQUIRE(graph, "%s: null graph", who);
  std::vector<hipGraphNode_t> nodes;
  if (int rc = list_nodes(who, (hipGraph_t)graph, nodes)) return rc;
  int nk = 0, ns = 0, nc = 0;
  for (hipGraphNode_t node : nodes) {
    hipGraphNodeType t;
    BG_HIP(hipGraphNodeGetType(node, &t), "hipGraphNodeGetType");
    nk += t == hipGraphNodeTypeKernel;
    ns += t == hipGraphNodeTypeMemset;
    nc += t == hipGraphNodeTypeMemcpy;
  }
  if (n_nodes) *n_nodes = (int)nodes.size();
  if (n_kernel) *n_kernel = nk;
  if (n_memset) *n_memset = ns;
  if (n_memcpy) *n_memcpy = nc;
  return BG_OK;
}

extern "C" int bg_graph_replace_memset_nodes(void* graph_, int* n_replaced) {
  const char* who = "bg_graph_replace_memset_nodes";
  BG_REQUIRE(graph_, "%s: null graph", who);
  hipGraph_t graph = (hipGraph_t)graph_;
  std::vector<hipGraphNode_t> nodes;
  if (int rc = list_nodes(who, graph, nodes)) return rc;
  int replaced = 0;
  for (hipGraphNode_t node : nodes) {
    hipGraphNodeType t;
    BG_HIP(hipGraphNodeGetType(node, &t), "hipGraphNodeGetType");
    if (t != hipGraphNodeTypeMemset) continue;
    hipMemsetParams mp;
    BG_HIP(hipGraphMemsetNodeGetParams(node, &mp), "hipGraphMemsetNodeGetParams");
    BG_REQUIRE(mp.elementSize == 1 || mp.elementSize == 2 || mp.elementSize == 4, "%s: memset node with element size %u", who, mp.elementSize);
    size_t n_in = 0, n_out = 0;
    BG_HIP(hipGraphNodeGetDependencies(node, nullptr, &n_in), "hipGraphNodeGetDependencies");
    std::vector<hipGraphNode_t> in(n_in);
    if (n_in) BG_HIP(hipGraphNodeGetDependencies(node, in.data(), &n_in), "hipGraphNodeGetDependencies");
    BG_HIP(hipGraphNodeGetDependentNodes(node, nullptr, &n_out), "hipGraphNodeGetDependentNodes");
    std::vector<hipGraphNode_t> outn(n_out);
    if (n_out) BG_HIP(hipGraphNodeGetDependentNodes(node, outn.data(), &n_out), "hipGraphNodeGetDependentNodes");

    unsigned char* dst = (unsigned char*)mp.dst;
    unsigned value = mp.value, esz = mp.elementSize;
    size_t width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
    if (width == 0) width = 1;            // an empty memset still orders its neighbours; the kernel writes nothing beyond `width`
    const bool empty = mp.width == 0;
    if (empty) height = 0;
    void* args[] = {&dst, &value, &esz, &width, &height, &pitch};
    hipKernelNodeParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.func = (void*)graph_fill_kernel;
    kp.blockDim = dim3(256);
    kp.gridDim = dim3((unsigned)cdiv((int64_t)width, 256), (unsigned)(height ? height : 1));
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    BG_REQUIRE(kp.gridDim.y <= 65535, "%s: memset node of %zu rows", who, height);
    hipGraphNode_t fresh;
    BG_HIP(hipGraphAddKernelNode(&fresh, graph, n_in ? in.data() : nullptr, n_in, &kp), "hipGraphAddKernelNode");
    for (size_t i = 0; i < n_out; ++i) BG_HIP(hipGraphAddDependencies(graph, &fresh, &outn[i], 1), "hipGraphAddDependencies");
    BG_HIP(hipGraphDestroyNode(node), "hipGraphDestroyNode");
    ++replaced;
  }
  if (n_replaced) *n_replaced = replaced;
  return BG_OK;
}
