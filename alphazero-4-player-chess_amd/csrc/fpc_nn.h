// fpc_nn.h -- placeholder until the MFMA ResNet lands (next commit)
#pragma once
#include <string>
#include "fpc_tree_kernels.h"
namespace fpc {
struct NN {
  bool loaded = false;
  int init(const DevCfg &, int, int, hipStream_t, std::string *) { return 0; }
  void destroy() {}
  int load(const void *, uint64_t, std::string *err) { *err = "NN not built yet"; return FPC_EWEIGHTS; }
  int forward(int, std::string *) { return FPC_EWEIGHTS; }
  int forward_external(const float *, int, float *, float *, std::string *) { return FPC_EWEIGHTS; }
  uint16_t *input16() { return nullptr; }
  uint16_t one16() { return 0; }
  float *logits() { return nullptr; }
  float *value() { return nullptr; }
};
}  // namespace fpc
