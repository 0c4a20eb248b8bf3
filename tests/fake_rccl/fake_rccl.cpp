// TEST INFRASTRUCTURE: an in-process stand-in for the few RCCL calls of pathtracer_amd/csrc/multi_gpu.cpp
// (ncclCommInitAll, ncclGroupStart / ncclGroupEnd, ncclSend, ncclRecv, ncclCommDestroy), so that its
// several-GPU path (one host thread per logical rank, the point-to-point gather on rank 0, the meeting
// before the exchange) can be run on ONE physical GPU, which real RCCL refuses. A send and its matching
// receive meet in a mailbox keyed by (from, to); the data moves with a device-to-device hipMemcpyAsync on the
// receiver's stream, ordered after the sender's stream by an event, and the sender's stream is ordered after
// the copy by a second event, which is what RCCL's stream semantics promise the caller. Built only into
// tests/_build/libptamd_multi_fake.so.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace {

struct Msg {
  const void *src = nullptr;
  size_t bytes = 0;
  hipEvent_t ready = nullptr;  // recorded on the sender's stream when the data is final
  hipEvent_t done = nullptr;   // recorded on the receiver's stream after the copy
  bool copied = false, failed = false;
};
struct World {
  int n = 0, alive = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<Msg *>> box;
};
struct Comm {
  World *w;
  int rank;
};
struct Op {
  bool send;
  void *buf;
  size_t bytes;
  int peer;
  Comm *c;
  hipStream_t s;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

// posts every send of the group, serves every receive, then waits for the sends to be taken
ncclResult_t run(std::vector<Op> &ops) {
  std::vector<std::pair<Op *, Msg *>> sent;
  for (Op &o : ops) {
    if (!o.send) continue;
    Msg *m = new Msg;
    m->src = o.buf;
    m->bytes = o.bytes;
    if (hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(m->ready, o.s) != hipSuccess) { delete m; return ncclUnhandledCudaError; }
    {
      std::lock_guard<std::mutex> lk(o.c->w->mu);
      o.c->w->box[{o.c->rank, o.peer}].push_back(m);
    }
    o.c->w->cv.notify_all();
    sent.push_back({&o, m});
  }
  ncclResult_t rc = ncclSuccess;
  for (Op &o : ops) {
    if (o.send) continue;
    World *w = o.c->w;
    Msg *m = nullptr;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      auto &q = w->box[{o.peer, o.c->rank}];
      w->cv.wait(lk, [&] { return !q.empty(); });
      m = q.front();
      q.pop_front();
    }
    bool ok = m->bytes == o.bytes;
    ok = ok && hipStreamWaitEvent(o.s, m->ready, 0) == hipSuccess;
    ok = ok && (o.bytes == 0 || hipMemcpyAsync(o.buf, m->src, o.bytes, hipMemcpyDeviceToDevice, o.s) == hipSuccess);
    hipEvent_t done = nullptr;
    ok = ok && hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess && hipEventRecord(done, o.s) == hipSuccess;
    {
      std::lock_guard<std::mutex> lk(w->mu);
      m->done = done;
      m->copied = true;
      m->failed = !ok;
    }
    w->cv.notify_all();
    if (!ok) rc = ncclInvalidArgument;
  }
  for (auto &sm : sent) {
    World *w = sm.first->c->w;
    Msg *m = sm.second;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return m->copied; });
    }
    if (m->failed) rc = ncclInvalidArgument;
    if (m->done) { (void)hipStreamWaitEvent(sm.first->s, m->done, 0); (void)hipEventDestroy(m->done); }
    (void)hipEventDestroy(m->ready);
    delete m;
  }
  return rc;
}

ncclResult_t enqueue(bool send, const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t s) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  const size_t tb = type_bytes(type);
  if (!c || !tb || peer < 0 || peer >= c->w->n || peer == c->rank) return ncclInvalidArgument;
  g_ops.push_back(Op{send, const_cast<void *>(buf), count * tb, peer, c, s});
  if (g_depth == 0) {
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run(ops);
  }
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
  (void)devlist;
  if (!comms || ndev < 1) return ncclInvalidArgument;
  World *w = new World;
  w->n = w->alive = ndev;
  for (int r = 0; r < ndev; r++) comms[r] = reinterpret_cast<ncclComm_t>(new Comm{w, r});
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c) return ncclSuccess;
  World *w = c->w;
  bool last;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    last = --w->alive == 0;
  }
  delete c;
  if (last) delete w;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  g_depth++;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run(ops);
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return enqueue(true, sendbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return enqueue(false, recvbuff, count, datatype, peer, comm, stream);
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "error reported by the in-process stand-in for RCCL"; }

}  // extern "C"
