/* jur_cli.c -- `formod <ctl> <obs> <atm> <rad> [KEY VALUE]...` on top of libjurassic_hip.so.
 *
 * The reference's forward-model executable (formod.c:33-69) so that its example scripts can call this
 * library's formod() unchanged; text I/O in jur_textio.c.  Everything numerical happens behind formod().
 */
#include <stdio.h>
#include <stdlib.h>
#include "jurassic_hip.h"

#include "jur_textio.h"

int main(int argc, char *argv[]) {
  if (argc < 5) DIE("Give parameters: <ctl> <obs> <atm> <rad>");
  ctl_t *ctl = (ctl_t *)calloc(1, sizeof(ctl_t));
  atm_t *atm = (atm_t *)calloc(1, sizeof(atm_t));
  obs_t *obs = (obs_t *)calloc(1, sizeof(obs_t));
  if (!ctl || !atm || !obs) DIE("Out of memory!");
  read_ctl(argc, argv, ctl);
  read_obs(argv[2], ctl, obs);
  read_atm(argv[3], ctl, atm);
  formod(ctl, atm, obs);
  write_obs(argv[4], ctl, obs);
  free(ctl); free(atm); free(obs);
  return EXIT_SUCCESS;
}
