/* multi_device.c -- a C caller of the multi-device entry (include/jurassic_hip.h: jur_formod_host_multi,
 * jur_formod_device_multi): the limb example's control block, a tangent-height scan IN ITS NATURAL ORDER (the case
 * equal ray counts per device would serve badly), one model per listed device.
 *   gcc -O2 -Iinclude tools/multi_device.c -Ljurassic-gpu_amd -ljurassic_hip -Wl,-rpath,$PWD/jurassic-gpu_amd -lm
 *   ./a.out <rays> <device> [<device> ...]        (tables ./boxcar_* and atm.tab of the limb example in cwd;
 *                                                   the same device may be listed several times: a rehearsal)
 * Prints one JSON line: the shares, the estimated and the actual LOS points per share, seconds per call with the
 * models listed against one model alone, and how many result values differ from the single-model call (0). */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "jurassic_hip.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CHECK(call) do { if ((call) != JUR_OK) { printf("%s failed: %s\n", #call, jur_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
  if (argc < 3) { printf("usage: multi_device <rays> <device> [<device> ...]\n"); return 2; }
  long const nr = atol(argv[1]);
  int const nm = argc - 2;
  if (nr < 1 || nm > 16) { printf("1 .. 16 devices, >= 1 ray\n"); return 2; }
  ctl_t *ctl = calloc(1, sizeof *ctl);
  atm_t *atm = calloc(1, sizeof *atm);
  char const *em[5] = {"CO2", "H2O", "O3", "F11", "CCl4"};
  ctl->ng = 5; ctl->nd = 2; ctl->nw = 1; ctl->nu[0] = 792; ctl->nu[1] = 832;
  for (int g = 0; g < 5; g++) strcpy(ctl->emitter[g], em[g]);
  strcpy(ctl->tblbase, "./boxcar"); ctl->hydz = -999; ctl->ctm_co2 = ctl->ctm_h2o = 1; ctl->ip = 1; ctl->refrac = 1;
  ctl->rayds = 10; ctl->raydz = 0.5; ctl->formod = 2; ctl->useGPU = 1;
  FILE *in = fopen("atm.tab", "r");
  if (!in) { printf("need atm.tab in cwd\n"); return 1; }
  char line[5000];
  while (fgets(line, sizeof line, in)) {
    int const i = atm->np;
    if (sscanf(line, "%lg %lg %lg %lg %lg %lg %lg %lg %lg %lg %lg %lg", &atm->time[i], &atm->z[i], &atm->lon[i], &atm->lat[i],
               &atm->p[i], &atm->t[i], &atm->q[0][i], &atm->q[1][i], &atm->q[2][i], &atm->q[3][i], &atm->q[4][i], &atm->k[0][i]) == 12)
      atm->np++;
  }
  fclose(in);

  jur_model_t *models[16], *single = NULL;
  for (int k = 0; k < nm; k++) CHECK(jur_model_create_from_files(&models[k], ctl, atoi(argv[2 + k])));
  CHECK(jur_model_create_from_files(&single, ctl, atoi(argv[2])));
  CHECK(jur_models_set_atm(models, nm, atm));
  CHECK(jur_model_set_atm(single, atm));

  /* the scan: tangent altitudes 3 .. 68 km, ascending (limb.c:49-59), pinned arrays */
  int const nd = ctl->nd;
  double *geom[7], *tp[3], *tp1[3];
  for (int k = 0; k < 7; k++) { geom[k] = jur_host_alloc(sizeof(double) * nr); memset(geom[k], 0, sizeof(double) * nr); }
  for (int k = 0; k < 3; k++) { tp[k] = jur_host_alloc(sizeof(double) * nr); tp1[k] = jur_host_alloc(sizeof(double) * nr); }
  double *rad = jur_host_alloc(sizeof(double) * nr * nd), *tau = jur_host_alloc(sizeof(double) * nr * nd);
  double *rad1 = jur_host_alloc(sizeof(double) * nr * nd), *tau1 = jur_host_alloc(sizeof(double) * nr * nd);
  int *np = malloc(sizeof(int) * nr), *np1 = malloc(sizeof(int) * nr);
  for (long i = 0; i < nr; i++) {
    double const z = 3 + 65.0 * i / (nr > 1 ? nr - 1 : 1);
    geom[1][i] = 780; geom[4][i] = z; geom[6][i] = 180 / M_PI * acos((JUR_RE + z) / (JUR_RE + 780));
  }
  double const *const cg[7] = {geom[0], geom[1], geom[2], geom[3], geom[4], geom[5], geom[6]};
  long bounds[17];
  CHECK(jur_multi_balance(models[0], nr, cg, nm, bounds));

  memset(rad1, 0, sizeof(double) * nr * nd);
  CHECK(jur_formod_host(single, nr, cg, rad1, tau1, tp1, np1));              /* warm-up and reference */
  double t0 = now();
  memset(rad1, 0, sizeof(double) * nr * nd);
  CHECK(jur_formod_host(single, nr, cg, rad1, tau1, tp1, np1));
  double const t_single = now() - t0;
  memset(rad, 0, sizeof(double) * nr * nd);
  CHECK(jur_formod_host_multi(models, nm, nr, cg, rad, tau, tp, np));
  t0 = now();
  memset(rad, 0, sizeof(double) * nr * nd);
  CHECK(jur_formod_host_multi(models, nm, nr, cg, rad, tau, tp, np));
  double const t_multi = now() - t0;

  long differing = 0;
  for (long i = 0; i < nr * nd; i++) differing += (memcmp(&rad[i], &rad1[i], 8) != 0) + (memcmp(&tau[i], &tau1[i], 8) != 0);
  for (long i = 0; i < nr; i++) {
    differing += np[i] != np1[i];
    for (int k = 0; k < 3; k++) differing += memcmp(&tp[k][i], &tp1[k][i], 8) != 0;
  }
  printf("{\"rays\": %ld, \"models\": %d, \"devices\": [", nr, nm);
  for (int k = 0; k < nm; k++) printf("%s%d", k ? ", " : "", atoi(argv[2 + k]));
  printf("], \"share_rays\": [");
  for (int k = 0; k < nm; k++) printf("%s%ld", k ? ", " : "", bounds[k + 1] - bounds[k]);
  printf("], \"share_los_points\": [");
  for (int k = 0; k < nm; k++) {
    long s = 0;
    for (long i = bounds[k]; i < bounds[k + 1]; i++) s += np[i];
    printf("%s%ld", k ? ", " : "", s);
  }
  printf("], \"equal_count_los_points\": [");
  for (int k = 0; k < nm; k++) {
    long s = 0;
    for (long i = nr * k / nm; i < nr * (k + 1) / nm; i++) s += np[i];
    printf("%s%ld", k ? ", " : "", s);
  }
  printf("], \"seconds_one_model\": %.5f, \"seconds_multi\": %.5f, \"differing_values\": %ld}\n", t_single, t_multi, differing);
  for (int k = 0; k < nm; k++) jur_model_destroy(models[k]);
  jur_model_destroy(single);
  return differing ? 1 : 0;
}
