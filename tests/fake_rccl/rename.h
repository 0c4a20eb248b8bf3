// TEST BUILD ONLY (force-included by `make -C pathtracer_amd/csrc test_multi`): the RCCL entry points that
// pathtracer_amd/csrc/multi_gpu.cpp calls are renamed, in <rccl/rccl.h>'s declarations and in its calls alike,
// to the in-process stand-ins of tests/fake_rccl/fake_rccl.cpp, so that the test library neither links nor
// collides with a real RCCL (torch brings its own copy into the test process).
#pragma once
#define ncclCommInitAll ptfake_ncclCommInitAll
#define ncclCommDestroy ptfake_ncclCommDestroy
#define ncclGroupStart ptfake_ncclGroupStart
#define ncclGroupEnd ptfake_ncclGroupEnd
#define ncclSend ptfake_ncclSend
#define ncclRecv ptfake_ncclRecv
#define ncclGetErrorString ptfake_ncclGetErrorString
