/* nb_internal.h — shared between the C and HIP translation units of libnbody_hip.so */
#ifndef NB_INTERNAL_H
#define NB_INTERNAL_H

#include "nbody.h"

#ifdef __cplusplus
extern "C" {
#endif

/* printf-style setters for the thread-local text / code behind nb_last_error() / nb_last_error_code():
 * nb_set_error records NB_EINVAL (bad argument: by far the commonest cause), nb_fail the given code,
 * which it also returns so that call sites read `return nb_fail(NB_EIO, ...)`. */
void nb_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int  nb_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void nb_clear_error(void);

#ifdef __cplusplus
}
#endif
#endif
