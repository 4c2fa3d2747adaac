// tests/emul/file_collective.cpp -- TEST INFRASTRUCTURE ONLY (never loaded by the product on its own: it is named
// through FPC_RCCL_LIB by tests/test_comm_two_ranks_gpu.py).
//
// RCCL refuses two ranks on one GPU ("duplicate GPU"), and this pool hands out one-GPU boxes, so the engine's own
// episode-end exchange (fpc_comm_init / fpc_allgather_tuples: counts + capacities -> buffer growth -> status round ->
// max-padded payload, csrc/fpc_engine.cpp) has only ever run with a one-rank communicator.  This stand-in exports the
// librccl entry points the engine binds with dlsym (the five it needs + ncclCommAbort) and implements ncclAllGather for N PROCESSES ON ONE BOX by way
// of files in a directory both ranks know (FILE_COLLECTIVE_DIR): every rank copies its send buffer device -> host,
// publishes it as <dir>/<seq>_<rank>.bin (write + rename), waits for every rank's file of that sequence number, and
// copies them host -> device into the receive buffer in rank order.  Blocking, like the real collective from the
// caller's point of view (the engine synchronises the stream behind each one anyway).  It says nothing about RCCL
// itself; it makes the ENGINE'S protocol around the collectives testable with more than one rank.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Comm {
  std::string dir;
  int rank = 0, world = 1;
  long seq = 0;
};
size_t dtype_size(int dt) {   // rccl.h ncclDataType_t: 0 int8, 1 uint8, 2 int32, 3 uint32, 4 int64, 5 uint64, 6 half, 7 float, 8 double
  switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: return 2; default: return 0; }
}
bool read_file(const std::string &p, std::vector<char> &buf, size_t want) {
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return false;
  buf.resize(want);
  const size_t got = fread(buf.data(), 1, want, f);
  fclose(f);
  return got == want;
}
}  // namespace

extern "C" {

int ncclGetUniqueId(void *id128) {
  memset(id128, 0, 128);
  const char *dir = getenv("FILE_COLLECTIVE_DIR");
  if (!dir || strlen(dir) > 120) return 5;      // ncclInvalidUsage
  strcpy((char *)id128, dir);
  return 0;
}

struct Id128 { char b[128]; };
int ncclCommInitRank(void **comm, int world, Id128 id, int rank) {
  Comm *c = new Comm();
  c->dir = std::string(id.b, strnlen(id.b, 128));
  c->rank = rank; c->world = world;
  *comm = c;
  return 0;
}

int ncclCommDestroy(void *comm) { delete (Comm *)comm; return 0; }
// ncclCommAbort: the rank leaves; a marker file makes every peer's current or next wait return ncclRemoteError (6)
// instead of running into its deadline -- the one property of the real call the engine relies on
int ncclCommAbort(void *comm) {
  Comm *c = (Comm *)comm;
  if (c) {
    if (FILE *f = fopen((c->dir + "/aborted").c_str(), "wb")) fclose(f);
    delete c;
  }
  return 0;
}
const char *ncclGetErrorString(int rc) { return rc == 0 ? "ok" : rc == 5 ? "invalid usage" : rc == 1 ? "unhandled hip error" : rc == 6 ? "remote error: a peer aborted its communicator" : "file collective: timeout or i/o error"; }

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) {
  Comm *c = (Comm *)comm;
  const size_t bytes = count * dtype_size(dtype);
  if (!c || !bytes) return 5;
  std::vector<char> mine(bytes);
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (hipMemcpy(mine.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const long seq = c->seq++;
  char name[64];
  snprintf(name, sizeof(name), "/%ld_%d.bin", seq, c->rank);
  const std::string fin = c->dir + name, tmp = fin + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return 3;
  const bool ok = fwrite(mine.data(), 1, bytes, f) == bytes;
  fclose(f);
  if (!ok || rename(tmp.c_str(), fin.c_str()) != 0) return 3;
  std::vector<char> other;
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(60);   // a rank that never arrives must not hang the test
  for (int r = 0; r < c->world; ++r) {
    snprintf(name, sizeof(name), "/%ld_%d.bin", seq, r);
    const std::string p = c->dir + name;
    const char *src = mine.data();
    if (r != c->rank) {
      while (!read_file(p, other, bytes)) {
        if (FILE *a = fopen((c->dir + "/aborted").c_str(), "rb")) { fclose(a); return 6; }
        if (std::chrono::steady_clock::now() > deadline) return 3;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
      src = other.data();
    }
    if (hipMemcpy((char *)recv + (size_t)r * bytes, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
  }
  return 0;
}

}  // extern "C"
