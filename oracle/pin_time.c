/* TEST INFRASTRUCTURE.  Linked into oracle/_ref/relateHMM_ref (the reference CLI built from its own sources) so
 * that its srand(time(0)) per pair (src/relateHMM.cpp:1853) is reproducible: time() returns $RELATE_PIN_TIME
 * (default 4242).  Not part of the product. */
#include <stdlib.h>
#include <time.h>
time_t time(time_t *t) {
  const char *e = getenv("RELATE_PIN_TIME");
  time_t v = e ? (time_t)atol(e) : (time_t)4242;
  if (t) *t = v;
  return v;
}
