// Weight-gradient side stream (see side.hip): an internal low-priority HIP stream per caller stream, forked from and
// joined back into the caller's stream with events inside one backward call.
#pragma once
#include "dcpt_common.h"

struct Side;
// nullptr when disabled (dcpt_set_side_stream(0), DCPT_SIDE_STREAM=0) or while `main` is being captured into a graph
Side* side_for(hipStream_t main);
hipStream_t side_stream(Side* sd, hipStream_t main);   // the side stream, or `main` when sd == nullptr
constexpr int SIDE_EVENTS = 8;
// work enqueued on `main` so far is visible to the side stream's next launches (event slot i < SIDE_EVENTS - 1)
int side_fork(Side* sd, int i, hipStream_t main);
// `main` continues only after everything enqueued on the side stream so far
int side_join(Side* sd, hipStream_t main);
