// Build-time tuning of the traversal kernels and their launch shape, shared by the kernels
// (trace_kernels.hip) and the host code that sizes their launches (ptamd_device.hip).
#pragma once

// Levels of a lane's traversal stack held in LDS (see TravStack in walk.h)
#ifndef PTAMD_STACK_EXTEND
#define PTAMD_STACK_EXTEND 14  // 5 workgroups per CU x 30 KiB
#endif
#ifndef PTAMD_STACK_SHADOW
#define PTAMD_STACK_SHADOW 10  // 7 workgroups per CU x 22 KiB
#endif
// Build-time tuning of the traversal kernels (defaults are the measured best): workgroups
// launched per resident workgroup slot by the grid-stride driver (REFILL == 0).
#ifndef PTAMD_GRID_MULT
#define PTAMD_GRID_MULT 8
#endif
#ifndef PTAMD_IN_ORDER
#define PTAMD_IN_ORDER 1  // first-hit walks take a node's children in the node's order (see node_step)
#endif
#ifndef PTAMD_NODE_STEPS
#define PTAMD_NODE_STEPS 2
#endif
#ifndef PTAMD_NODE_TAGS
#define PTAMD_NODE_TAGS 1  // end primitives are rejected in the box step of the node that holds them (see node_step)
#endif
// idle lanes at which a persistent wave refills (0: grid-stride driver), see trace_rays
#ifndef PTAMD_REFILL_EXTEND
#define PTAMD_REFILL_EXTEND 24
#endif
// Shadow rays: 40 was best while the lists still held the rays that the light's own quad occludes; swept
// again without them, 24 is 1 % faster per frame on the 100k-triangle scene and equal on CBbunny, 16 no
// better there and 0.6 % slower on CBbunny, 8 and 48 slower on both.
#ifndef PTAMD_REFILL_SHADOW
#define PTAMD_REFILL_SHADOW 24
#endif
#ifndef PTAMD_REFILL_TWO_S
#define PTAMD_REFILL_TWO_S 24  // the shadow-ray list of k_trace2
#endif
#ifndef PTAMD_WAVES_EXTEND
#define PTAMD_WAVES_EXTEND 5  // the closest-hit walk with lane refill needs ~96 registers; spilling costs more than a wave
#endif
#ifndef PTAMD_WAVES_SHADOW
#define PTAMD_WAVES_SHADOW 7  // first-hit walks: 72 registers, 4 of them spilled; 8 waves lose
#endif
#define TRAV_BOUNDS_N(n) __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(n, n)))
constexpr int TRAV_BLOCK = 256;
