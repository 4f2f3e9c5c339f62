// Placement arithmetic of the stream calls, as pure functions of the PCM's counters.
//
// SoapySX::readStream / writeStream (SoapySX.cpp:868-1105) decide three things from a handful of integers
// before any sample moves: how far a capture stream must jump after an overrun, where a playback block lands
// (at its timestamp, right behind the previous block, or past an underrun), and how much of a request a
// non-blocking call may take.  They are restated here without any I/O so that the Device (SoapySXHip.cpp) is
// a thin sequence of "ask the rule, move the PCM", and so that the rules can be swept on the CPU against the
// oracle's independent restatement (tests/host/host_logic_probe.cpp, tests/test_host_logic.py).
#pragma once

#include <cstdint>

namespace sx {
namespace rules {

// Both xrun recoveries use the same catch-up distance: the whole periods that were lost plus a margin of one
// to two periods, so that the stream lands on a period boundary safely past the hole
// (capture: SoapySX.cpp:910-916, playback: SoapySX.cpp:1030-1033).
inline int64_t catch_up(int64_t lost, uint64_t period)
{
    return (lost / (int64_t)period + 2) * (int64_t)period;
}

// Capture overrun: more frames available than the ring holds means the oldest were overwritten.  Returns the
// number of frames to forward the stream by (0 = no overrun).  SoapySX.cpp:910-927.
inline int64_t rx_overrun_skip(int64_t avail, uint64_t ring, uint64_t period)
{
    return avail > (int64_t)ring ? catch_up(avail - (int64_t)ring, period) : 0;
}

// How much of a request a call takes: everything when it may block, what is there (or fits) now when
// timeoutUs <= 0.  SoapySX.cpp:934-942 (capture), :1076-1085 (playback).
inline uint64_t request_length(uint64_t wanted, int64_t avail, long timeout_us)
{
    if (timeout_us > 0) return wanted;
    if (avail <= 0) return 0;
    return (uint64_t)avail < wanted ? (uint64_t)avail : wanted;
}

// Where a playback block goes.  SoapySX.cpp:1000-1038.
struct TxPlacement {
    enum Kind {
        IN_SEQUENCE,     // untimed, the device has not run past the stream position: right behind the last block
        AT_TIMESTAMP,    // timed, not in the past: at the position its timestamp names (a gap plays as silence)
        PAST_UNDERRUN,   // untimed, the device ran dry `late` frames ago: forward by catch_up(late)
        IN_THE_PAST      // timed, already played: dropped, but reported as written
    } kind;
    int64_t write_position;   // stream position of the block's first sample (undefined for IN_THE_PAST)
    int64_t late;             // frames by which the playback position is ahead of the requested position (> 0: late)
};

// position: stream position of the next sample the application would write; delay: frames between that and the
// sample being played (playback_position = position - delay); timed_position: the block's timestamp in frames
// (only looked at when `timed`).
inline TxPlacement tx_placement(int64_t position, int64_t delay, uint64_t period, bool timed, int64_t timed_position)
{
    const int64_t playing = position - delay;
    TxPlacement p;
    if (timed) {
        p.late = playing - timed_position;
        p.kind = p.late > 0 ? TxPlacement::IN_THE_PAST : TxPlacement::AT_TIMESTAMP;
        p.write_position = timed_position;
    } else {
        p.late = playing - position;
        p.kind = p.late > 0 ? TxPlacement::PAST_UNDERRUN : TxPlacement::IN_SEQUENCE;
        p.write_position = p.late > 0 ? position + catch_up(p.late, period) : position;
    }
    return p;
}

}  // namespace rules
}  // namespace sx
