/* nb_internal.h — shared between the C and HIP translation units of libnbody_hip.so */
#ifndef NB_INTERNAL_H
#define NB_INTERNAL_H

#include "nbody.h"

#ifdef __cplusplus
extern "C" {
#endif

/* printf-style setter for the thread-local text behind nb_last_error() */
void nb_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void nb_clear_error(void);

#ifdef __cplusplus
}
#endif
#endif
