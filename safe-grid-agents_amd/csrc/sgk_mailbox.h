// sgk_mailbox.h -- the single-env step server's mailbox: the one data structure the device side (env_server_kernel, sgk_step.hip) and
// the host side (stop_server / server_round_trip, sgk_host_core.h) of the protocol share. Plain C layout, no HIP types.
#pragma once
#include <stdint.h>

namespace sgk {

// Pinned, device-mapped host memory; each word on a cache line of its own.
struct SgkMailbox {
  // host -> device, ONE 8-byte word so that one PCIe read carries the whole request of a single env: bits 0..31 the number of
  // the step asked for (never SGK_SERVER_STOP; SGK_SERVER_STOP = leave), bits 32..39 the SGK_F_* flags of that step, bits 40..47
  // env 0's action (the other envs' actions, if any, are read from the host-visible action buffer)
  volatile uint64_t request;
  uint32_t pad0[14];
  volatile uint32_t done;     // device -> host: number of the last step whose outputs are in the host-visible buffers
  uint32_t pad2[15];
  volatile uint32_t exited;   // device -> host: 0 while the server runs; (last step served + 1) once it has left
  uint32_t pad3[15];
};

}  // namespace sgk

#define SGK_SERVER_STOP 0xffffffffu
#define SGK_SRV_RESET 0x80u     // in the flags byte of a step-server request: reset every env of the handle instead of stepping
#define SGK_SERVER_IDLE_US 100  // the step server leaves after this long without a request (wall_clock64: 100 MHz)
