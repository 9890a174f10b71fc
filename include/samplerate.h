/*
 * samplerate.h -- drop-in for the nine C symbols LibRedio's Rust binds from libsamplerate
 * (src/samplerate/src/samplerate.rs:32-42, linked by name "samplerate"):
 *     src_new, src_delete, src_process, src_get_name, src_get_description, src_get_version,
 *     src_set_ratio, src_is_valid_ratio, src_strerror
 * plus src_reset / src_error / src_simple for C callers.  Exported by libsamplerate.so in this repo;
 * the arithmetic runs on the MI355X through redio_src_* (include/redio.h).
 *
 * SRC_DATA is the C layout behind the Rust struct at samplerate.rs:15-24 (64 bytes on LP64).
 * src_process writes input_frames_used / output_frames_gen back through the pointer, as the
 * reference relies on (samplerate.rs:76,84).  Converter 1 (SRC_SINC_MEDIUM_QUALITY), channels 1 is
 * what the reference uses (:61); all five converters it declares (:26-30: three sinc classes, zero-order
 * hold, linear) and channels >= 1 (interleaved frames, each channel converted as a mono stream) are built;
 * a converter above 4 or a channel count below 1 returns NULL with the library's error code.  Deviation stated in DESIGN.md: the coefficient tables are not the
 * library's (they cannot be reproduced here), so sample values differ from the real library within
 * its quality class; control flow and frame counts follow the published 0.1.8 algorithm for ONE channel (what the
 * reference uses).  Four stated definitions beyond it: (i) with channels > 1 and a sinc converter every channel runs through
 * its own mono state, so the frame count of the LAST message of a stream (end_of_input = 1) is the mono loop's: the library's
 * multi-channel loops test their end-of-input condition in sample units with >= where the mono loop has >, and can differ from
 * this by one frame there; (ii) a message of a single frame through the zero-order-hold / linear converters interpolates from
 * the frame carried from the previous message (the published loop reads data_in[-channels] at that point); known answers:
 * tests/test_src_single_frame_kat.py; (iii) when src_ratio FALLS between two calls the sinc filter widens (half_filter_chan_len follows min(last_ratio, src_ratio)) while
 * b_current still sits where the narrower filter left it, so calc_output_single's left wing (data_index = b_current - coeff_count)
 * and prepare_data's move source (b_current - half_filter_chan_len) can be NEGATIVE: libsamplerate 0.1.8 reads the words in front of
 * its buffer there (the filter struct's own fields).  This library and its oracle read +0.0f, silence before the stream
 * (found by the randomised run of round 4; tests/test_gpu_resample.py::test_ratio_decrease_reaches_in_front_of_the_buffer);
 * (iv) with end_of_input = 1 and a ratio below about 1 / 213, prepare_data's last move (memmove of half_filter_chan_len + (b_end - b_current)
 * floats to the start of the buffer) can be LONGER than the buffer, which holds 2.5 half-lengths of the widest filter: libsamplerate 0.1.8
 * writes past its allocation there and then zero-fills a negative length.  This library and its oracle return
 * SRC_ERR_SINC_PREPARE_DATA_BAD_LEN (the library's own code for an impossible length) from that call instead, with
 * input_frames_used = output_frames_gen = 0 (found by the randomised run of round 4, ratios down to 1 / 256;
 * tests/test_gpu_resample.py::test_end_of_input_at_the_smallest_ratios).  The reference never sets end_of_input (samplerate.rs:73).
 */
#ifndef SAMPLERATE_H
#define SAMPLERATE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct SRC_STATE_tag SRC_STATE;

typedef struct {
    const float *data_in;
    float *data_out;
    long input_frames, output_frames;
    long input_frames_used, output_frames_gen;
    int end_of_input;
    double src_ratio;
} SRC_DATA;

enum {
    SRC_SINC_BEST_QUALITY = 0,   /* samplerate.rs:26 */
    SRC_SINC_MEDIUM_QUALITY = 1, /* :27 */
    SRC_SINC_FASTEST = 2,        /* :28 */
    SRC_ZERO_ORDER_HOLD = 3,     /* :29 */
    SRC_LINEAR = 4               /* :30 */
};

SRC_STATE *src_new(int converter_type, int channels, int *error);
SRC_STATE *src_delete(SRC_STATE *state);
int src_process(SRC_STATE *state, SRC_DATA *data);
const char *src_get_name(int converter_type);
const char *src_get_description(int converter_type);
const char *src_get_version(void);
int src_set_ratio(SRC_STATE *state, double new_ratio);
int src_is_valid_ratio(double ratio);
const char *src_strerror(int error);
int src_reset(SRC_STATE *state);
int src_error(SRC_STATE *state);
int src_simple(SRC_DATA *data, int converter_type, int channels);

#ifdef __cplusplus
}
#endif
#endif
