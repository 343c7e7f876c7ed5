/* lanes_bench.c -- throughput of the drop-in formod() with concurrent callers (the reference's usage:
 * OpenMP threads, each with its own obs package of <= 1088 rays; GPUdrivers.cu:262-342).
 *   gcc -O2 -fopenmp -Iinclude tools/lanes_bench.c -Ljurassic-gpu_amd -ljurassic_hip -Wl,-rpath,$PWD/jurassic-gpu_amd -lm
 *   JUR_LANES=8 ./a.out <ctl-dir with tables> <nthreads> <calls per thread>
 * The control block is the limb example's (5 emitters, channels 792/832); tables ./boxcar_* in cwd. */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jurassic_hip.h"

static double now(void) { return omp_get_wtime(); }

int main(int argc, char **argv) {
  int const nthr = argc > 1 ? atoi(argv[1]) : 4, ncall = argc > 2 ? atoi(argv[2]) : 8;
  ctl_t *ctl = calloc(1, sizeof *ctl);
  atm_t *atm = calloc(1, sizeof *atm);
  char const *em[5] = {"CO2", "H2O", "O3", "F11", "CCl4"};
  ctl->ng = 5; ctl->nd = 2; ctl->nw = 1; ctl->nu[0] = 792; ctl->nu[1] = 832;
  for (int g = 0; g < 5; g++) strcpy(ctl->emitter[g], em[g]);
  strcpy(ctl->tblbase, "./boxcar"); ctl->hydz = -999; ctl->ctm_co2 = ctl->ctm_h2o = 1; ctl->ip = 1; ctl->refrac = 1;
  ctl->rayds = 10; ctl->raydz = 0.5; ctl->formod = 2; ctl->useGPU = 1;
  /* atmosphere: atm.tab of the limb example in cwd */
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
  obs_t **obs = malloc(sizeof(obs_t *) * nthr);
  for (int t = 0; t < nthr; t++) {
    obs[t] = calloc(1, sizeof(obs_t));
    obs[t]->nr = JUR_NR;
    for (int i = 0; i < JUR_NR; i++) {
      double const z = 3 + 65.0 * ((i * 7919 + t * 104729) % JUR_NR) / JUR_NR;
      obs[t]->obsz[i] = 780; obs[t]->vpz[i] = z;
      obs[t]->vplat[i] = 180 / M_PI * acos((JUR_RE + z) / (JUR_RE + 780));
    }
  }
  formod(ctl, atm, obs[0]);                                   /* loads the tables */
#pragma omp parallel for num_threads(nthr) schedule(static, 1)
  for (int t = 0; t < nthr; t++) formod(ctl, atm, obs[t]);    /* creates the lanes (streams, staging buffers): once per process */
  double const t0 = now();
#pragma omp parallel for num_threads(nthr) schedule(static, 1)
  for (int t = 0; t < nthr; t++)
    for (int c = 0; c < ncall; c++) formod(ctl, atm, obs[t]);
  double const dt = now() - t0;
  printf("{\"threads\": %d, \"calls\": %d, \"rays_per_call\": %d, \"seconds\": %.4f, \"rays_per_s\": %.0f, \"rad0\": %.10g}\n", nthr,
         nthr * ncall, JUR_NR, dt, (double)nthr * ncall * JUR_NR / dt, obs[0]->rad[0][0]);
  return 0;
}
