/* jur_textio.h -- the reference's text formats and control-file parser, shared by the command-line
 * tools (formod, limb, nadir, climatology).  Restated from src/jurassic.c: read_ctl :920-1021,
 * scan_ctl :1153-1201, read_atm :882-917, read_obs :1041-1068, write_atm :1249-1277,
 * write_obs :1426-1470.  Plain C, no GPU. */
#ifndef JUR_TEXTIO_H
#define JUR_TEXTIO_H
#include <stdio.h>
#include <stdlib.h>
#include "jurassic_abi.h"

#define DIE(...)                                                     \
  do {                                                               \
    printf("\nError (%s, l%d): ", __FILE__, __LINE__);               \
    printf(__VA_ARGS__);                                             \
    printf("\n\n");                                                  \
    exit(EXIT_FAILURE);                                              \
  } while (0)

double scan_ctl(int argc, char *argv[], char const *varname, int arridx, char const *defvalue, char *value);
void read_ctl(int argc, char *argv[], ctl_t *ctl);
void read_atm(char const *filename, ctl_t const *ctl, atm_t *atm);
void read_obs(char const *filename, ctl_t const *ctl, obs_t *obs);
void write_atm(char const *filename, ctl_t const *ctl, atm_t const *atm);
void write_obs(char const *filename, ctl_t const *ctl, obs_t const *obs);

#endif
