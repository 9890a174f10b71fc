// samplerate_shim.cpp -- libsamplerate.so: the C symbols of src/samplerate/src/samplerate.rs:32-42 on
// top of redio_src_*.  See include/samplerate.h for the contract.
#include "../../include/samplerate.h"
#include "../../include/redio.h"
#include <stdlib.h>

struct SRC_STATE_tag {
    redio_src *h;
    int last_error;
};

extern "C" SRC_STATE *src_new(int converter_type, int channels, int *error)
{
    if (error) *error = 0;
    if (channels < 1) { if (error) *error = REDIO_SRC_ERR_BAD_CHANNEL_COUNT; return NULL; }
    SRC_STATE *st = (SRC_STATE *)calloc(1, sizeof(SRC_STATE));
    if (!st) { if (error) *error = REDIO_SRC_ERR_MALLOC_FAILED; return NULL; }
    int rc = redio_src_create(&st->h, converter_type, channels); // the reference asks for (1, 1) (samplerate.rs:61)
    if (rc != REDIO_OK) {
        if (error) *error = rc > 0 ? rc : REDIO_SRC_ERR_MALLOC_FAILED; // a HIP failure has no libsamplerate code
        free(st);
        return NULL;
    }
    return st;
}

extern "C" SRC_STATE *src_delete(SRC_STATE *st)
{
    if (st) { redio_src_destroy(st->h); free(st); }
    return NULL;
}

extern "C" int src_process(SRC_STATE *st, SRC_DATA *d)
{
    if (!st) return REDIO_SRC_ERR_BAD_STATE;
    if (!d) return REDIO_SRC_ERR_BAD_DATA;
    long used = 0, gen = 0;
    int rc = redio_src_process_host(st->h, d->data_in, d->input_frames, d->data_out, d->output_frames, d->src_ratio,
                                    d->end_of_input, &used, &gen);
    d->input_frames_used = used;
    d->output_frames_gen = gen;
    st->last_error = rc;
    return rc;
}

extern "C" int src_set_ratio(SRC_STATE *st, double r) { return st ? redio_src_set_ratio(st->h, r) : REDIO_SRC_ERR_BAD_STATE; }
extern "C" int src_is_valid_ratio(double r) { return !(r < (1.0 / 256) || r > 256.0); }
extern "C" int src_reset(SRC_STATE *st) { return st ? redio_src_reset(st->h) : REDIO_SRC_ERR_BAD_STATE; }
extern "C" int src_error(SRC_STATE *st) { return st ? st->last_error : 0; }

extern "C" const char *src_get_name(int c)
{
    switch (c) {
    case 0: return "Best Sinc Interpolator";
    case 1: return "Medium Sinc Interpolator";
    case 2: return "Fastest Sinc Interpolator";
    case 3: return "ZOH Interpolator";
    case 4: return "Linear Interpolator";
    default: return NULL;
    }
}
extern "C" const char *src_get_description(int c)
{
    switch (c) {
    case 0: return "Band limited sinc interpolation, best quality class (MI355X, libredio table).";
    case 1: return "Band limited sinc interpolation, medium quality class (MI355X, libredio table).";
    case 2: return "Band limited sinc interpolation, fastest class (MI355X, libredio table).";
    case 3: return "Zero order hold interpolator, very fast, poor quality.";
    case 4: return "Linear interpolator, very fast, poor quality.";
    default: return NULL;
    }
}
extern "C" const char *src_get_version(void) { return "libredio-samplerate 0.1 (API of libsamplerate 0.1.8, gfx950)"; }

extern "C" const char *src_strerror(int error)
{
    switch (error) {
    case 0: return "No error.";
    case REDIO_SRC_ERR_MALLOC_FAILED: return "Malloc failed.";
    case REDIO_SRC_ERR_BAD_STATE: return "SRC_STATE pointer is NULL.";
    case REDIO_SRC_ERR_BAD_DATA: return "SRC_DATA pointer is NULL.";
    case REDIO_SRC_ERR_BAD_DATA_PTR: return "SRC_DATA->data_out or SRC_DATA->data_in is NULL.";
    case REDIO_SRC_ERR_BAD_SRC_RATIO: return "SRC ratio outside [1/256, 256] range.";
    case REDIO_SRC_ERR_BAD_CONVERTER: return "Bad converter number.";
    case REDIO_SRC_ERR_BAD_CHANNEL_COUNT: return "Channel count must be >= 1.";
    case REDIO_SRC_ERR_DATA_OVERLAP: return "Input and output data arrays overlap.";
    case REDIO_SRC_ERR_SINC_PREPARE_DATA_BAD_LEN: return "Internal error : Bad length in prepare_data ().";
    case REDIO_SRC_ERR_BAD_INTERNAL_STATE: return "Error : Someone is trampling on my internal state.";
    default: break;
    }
    if (error < 0) return redio_strerror(error);
    return NULL;
}

extern "C" int src_simple(SRC_DATA *d, int converter, int channels)
{
    int err = 0;
    SRC_STATE *st = src_new(converter, channels, &err);
    if (!st) return err;
    d->end_of_input = 1;
    err = src_process(st, d);
    src_delete(st);
    return err;
}
