/* jur_textio.c -- see jur_textio.h */
#define _GNU_SOURCE
#include <math.h>
#include <stddef.h>
#include <string.h>
#include <strings.h>
#include "jur_textio.h"

/* KEY = VALUE from the control file (first match), overridden by `KEY VALUE` pairs on the command
 * line; KEY[i] also matches KEY[*]; an empty default makes the key mandatory. */
double scan_ctl(int argc, char *argv[], char const *varname, int arridx, char const *defvalue, char *value) {
  static char line[JUR_LEN], rvarname[JUR_LEN], dummy[JUR_LEN], rval[JUR_LEN];
  char full1[JUR_LEN], full2[JUR_LEN];
  int contain = 0;
  if (arridx >= 0) {
    snprintf(full1, sizeof full1, "%s[%d]", varname, arridx);
    snprintf(full2, sizeof full2, "%s[*]", varname);
  } else {
    snprintf(full1, sizeof full1, "%s", varname);
    snprintf(full2, sizeof full2, "%s", varname);
  }
  if (argv[1][0] != '-') {
    FILE *in = fopen(argv[1], "r");
    if (!in) DIE("cannot open control file %s", argv[1]);
    while (fgets(line, JUR_LEN, in))
      if (sscanf(line, "%4999s %4999s %4999s", rvarname, dummy, rval) == 3)
        if (0 == strcasecmp(rvarname, full1) || 0 == strcasecmp(rvarname, full2)) {
          contain = 1;
          break;
        }
    fclose(in);
  }
  for (int i = 1; i < argc - 1; i++)
    if (0 == strcasecmp(argv[i], full1) || 0 == strcasecmp(argv[i], full2)) {
      snprintf(rval, sizeof rval, "%s", argv[i + 1]);
      contain = 1;
      break;
    }
  if (!contain) {
    if (strlen(defvalue) > 0) snprintf(rval, sizeof rval, "%s", defvalue);
    else DIE("Missing variable %s!", full1);
  }
  if (arridx < 0) printf("%s = %s\n", full1, rval);
  if (value) snprintf(value, JUR_LEN, "%s", rval);
  return atof(rval);
}

/* The control block is filled from a table of its keys: one row per key with the member it lands in, its type,
 * its default ("" = mandatory) and what it is an array over.  Keys, defaults and their order are the file format of
 * the reference's control files (read_ctl, jurassic.c:920-1022) -- `formod limb.ctl ...` must accept the same files. */
enum { T_INT, T_DBL, T_STR };
enum { PER_NONE, PER_GAS, PER_CHANNEL, PER_WINDOW };
typedef struct {
  char const *key;
  int type, per;
  size_t offset, stride;
  char const *dflt;
} ctl_key_t;
#define ROW(key, type, per, member, dflt) {key, type, per, offsetof(ctl_t, member), 0, dflt}
#define ROWS(key, type, per, member, dflt) {key, type, per, offsetof(ctl_t, member), sizeof(((ctl_t *)0)->member[0]), dflt}
static ctl_key_t const ctl_keys[] = {
  ROW("NG", T_INT, PER_NONE, ng, "0"),
  ROWS("EMITTER", T_STR, PER_GAS, emitter, ""),
  ROW("ND", T_INT, PER_NONE, nd, "0"),
  ROWS("NU", T_DBL, PER_CHANNEL, nu, ""),
  ROW("NW", T_INT, PER_NONE, nw, "1"),
  ROWS("WINDOW", T_INT, PER_CHANNEL, window, "0"),
  ROW("TBLBASE", T_STR, PER_NONE, tblbase, "-"),
  ROW("HYDZ", T_DBL, PER_NONE, hydz, "-999"),
  ROW("CTM_CO2", T_INT, PER_NONE, ctm_co2, "1"),
  ROW("CTM_H2O", T_INT, PER_NONE, ctm_h2o, "1"),
  ROW("CTM_N2", T_INT, PER_NONE, ctm_n2, "1"),
  ROW("CTM_O2", T_INT, PER_NONE, ctm_o2, "1"),
  ROW("IP", T_INT, PER_NONE, ip, "1"),
  ROW("CZ", T_DBL, PER_NONE, cz, "0"),
  ROW("CX", T_DBL, PER_NONE, cx, "0"),
  ROW("REFRAC", T_INT, PER_NONE, refrac, "1"),
  ROW("RAYDS", T_DBL, PER_NONE, rayds, "10"),
  ROW("RAYDZ", T_DBL, PER_NONE, raydz, "0.5"),
  ROW("FOV", T_STR, PER_NONE, fov, "-"),
  ROW("RETP_ZMIN", T_DBL, PER_NONE, retp_zmin, "-999"),
  ROW("RETP_ZMAX", T_DBL, PER_NONE, retp_zmax, "-999"),
  ROW("RETT_ZMIN", T_DBL, PER_NONE, rett_zmin, "-999"),
  ROW("RETT_ZMAX", T_DBL, PER_NONE, rett_zmax, "-999"),
  ROWS("RETQ_ZMIN", T_DBL, PER_GAS, retq_zmin, "-999"),
  ROWS("RETQ_ZMAX", T_DBL, PER_GAS, retq_zmax, "-999"),
  ROWS("RETK_ZMIN", T_DBL, PER_WINDOW, retk_zmin, "-999"),
  ROWS("RETK_ZMAX", T_DBL, PER_WINDOW, retk_zmax, "-999"),
  ROW("WRITE_BBT", T_INT, PER_NONE, write_bbt, "0"),
  ROW("WRITE_MATRIX", T_INT, PER_NONE, write_matrix, "0"),
  ROW("FORMOD", T_INT, PER_NONE, formod, "2"),
  ROW("RFMBIN", T_STR, PER_NONE, rfmbin, "-"),
  ROW("RFMHIT", T_STR, PER_NONE, rfmhit, "-"),
  ROWS("RFMXSC", T_STR, PER_GAS, rfmxsc, "-"),
  ROW("USEGPU", T_INT, PER_NONE, useGPU, "0"),
  ROW("CHECKMODE", T_INT, PER_NONE, checkmode, "0"),
  ROW("READ_BINARY", T_INT, PER_NONE, read_binary, "-1"),
  ROW("WRITE_BINARY", T_INT, PER_NONE, write_binary, "1"),
  ROW("GPU_SHARED_MEMORY", T_INT, PER_NONE, gpu_nbytes_shared_memory, "0"),
};
#undef ROW
#undef ROWS

/* A continuum is switched off when no channel lies in the spectral range its coefficients cover (the reference
 * does this while reading the control file, jurassic.c:954-968). */
static void drop_continua_without_channels(ctl_t *ctl) {
  struct { int *flag; char const *gas; double lo, hi; int hi_open; } const span[] = {
    {&ctl->ctm_co2, "CO2", -INFINITY, 4000, 1},      /* nu < 4000        (jr_common.h:318) */
    {&ctl->ctm_h2o, "H2O", -INFINITY, 20000, 1},     /* nu < 20000       (:345)            */
    {&ctl->ctm_n2, "N2", 2120, 2605, 0},             /* 2120 .. 2605     (:367)            */
    {&ctl->ctm_o2, "O2", 1360, 1805, 0},             /* 1360 .. 1805     (:381)            */
  };
  for (size_t k = 0; k < sizeof span / sizeof span[0]; k++) {
    int covered = 0;
    for (int id = 0; id < ctl->nd && !covered; id++)
      covered = ctl->nu[id] >= span[k].lo && (span[k].hi_open ? ctl->nu[id] < span[k].hi : ctl->nu[id] <= span[k].hi);
    if (!covered && *span[k].flag) {
      *span[k].flag = 0;
      printf("No frequency in %s range, automatically set CTM_%s = 0\n", span[k].gas, span[k].gas);
    }
  }
}

void read_ctl(int argc, char *argv[], ctl_t *ctl) {
  printf("\nJuelich Rapid Spectral Simulation Code (JURASSIC), MI355X forward model\n(executable: %s)\n\n", argv[0]);
  for (size_t k = 0; k < sizeof ctl_keys / sizeof ctl_keys[0]; k++) {
    ctl_key_t const *row = &ctl_keys[k];
    int const n = row->per == PER_GAS ? ctl->ng : row->per == PER_CHANNEL ? ctl->nd : row->per == PER_WINDOW ? ctl->nw : 1;
    for (int i = 0; i < n; i++) {
      char *const dst = (char *)ctl + row->offset + (size_t)i * row->stride;
      int const idx = row->per == PER_NONE ? -1 : i;
      if (row->type == T_STR) scan_ctl(argc, argv, row->key, idx, row->dflt, dst);
      else if (row->type == T_INT) *(int *)dst = (int)scan_ctl(argc, argv, row->key, idx, row->dflt, NULL);
      else *(double *)dst = scan_ctl(argc, argv, row->key, idx, row->dflt, NULL);
    }
    /* the counts bound the arrays that follow them */
    if (0 == strcmp(row->key, "NG") && (ctl->ng < 0 || ctl->ng > JUR_NG)) DIE("Set 0 <= NG <= %d", JUR_NG);
    if (0 == strcmp(row->key, "ND") && (ctl->nd < 0 || ctl->nd > JUR_ND)) DIE("Set 0 <= ND <= %d", JUR_ND);
    if (0 == strcmp(row->key, "NW") && (ctl->nw < 0 || ctl->nw > JUR_NW)) DIE("Set 0 <= NW <= %d", JUR_NW);
    if (0 == strcmp(row->key, "CTM_O2")) drop_continua_without_channels(ctl);
    if (0 == strcmp(row->key, "CHECKMODE"))
      printf("CHECKMODE = %d (%s)\n", ctl->checkmode, (0 == ctl->checkmode) ? "run" : ((ctl->checkmode > 0) ? "skip" : "obs"));
  }
}

/* one whitespace-separated number per call; a token that does not parse drops the whole line,
 * as the reference's TOK macro does (jurassic.h:95-99) */
static int next_number(char **save, char *first, double *out) {
  char *tok = strtok_r(first, " \t", save);   /* newline is not a separator upstream: a blank line is one unparsable token */
  if (!tok) DIE("Error while reading!");
  return sscanf(tok, "%lg", out) == 1;
}

void read_atm(char const *filename, ctl_t const *ctl, atm_t *atm) {
  static char line[JUR_LEN];
  atm->init = 0;
  atm->np = 0;
  printf("Read atmospheric data: %s\n", filename);
  FILE *in = fopen(filename, "r");
  if (!in) DIE("cannot open %s", filename);
  if (ctl->checkmode) { fclose(in); return; }
  while (fgets(line, JUR_LEN, in)) {
    char *save;
    int const i = atm->np;
    if (i >= JUR_NP) DIE("Too many data points!");
    if (!next_number(&save, line, &atm->time[i]) || !next_number(&save, NULL, &atm->z[i]) ||
        !next_number(&save, NULL, &atm->lon[i]) || !next_number(&save, NULL, &atm->lat[i]) ||
        !next_number(&save, NULL, &atm->p[i]) || !next_number(&save, NULL, &atm->t[i])) continue;
    int ok = 1;
    for (int ig = 0; ig < ctl->ng && ok; ig++) ok = next_number(&save, NULL, &atm->q[ig][i]);
    for (int iw = 0; iw < ctl->nw && ok; iw++) ok = next_number(&save, NULL, &atm->k[iw][i]);
    if (!ok) continue;
    atm->np++;
  }
  fclose(in);
  if (atm->np < 1) DIE("Could not read any data!");
  printf("Read atmospheric data found %d height levels, max %d\n", atm->np, JUR_NP);
}

void read_obs(char const *filename, ctl_t const *ctl, obs_t *obs) {
  static char line[JUR_LEN];
  obs->nr = 0;
  printf("Read observation data: %s\n", filename);
  FILE *in = fopen(filename, "r");
  if (!in) DIE("cannot open %s", filename);
  if (ctl->checkmode > 0) { fclose(in); return; }
  while (fgets(line, JUR_LEN, in)) {
    char *save;
    int const i = obs->nr;
    if (i >= JUR_NR) DIE("Too many rays!");
    double *col[10] = {&obs->time[i], &obs->obsz[i], &obs->obslon[i], &obs->obslat[i], &obs->vpz[i],
                       &obs->vplon[i], &obs->vplat[i], &obs->tpz[i], &obs->tplon[i], &obs->tplat[i]};
    int ok = next_number(&save, line, col[0]);
    for (int c = 1; c < 10 && ok; c++) ok = next_number(&save, NULL, col[c]);
    for (int id = 0; id < ctl->nd && ok; id++) ok = next_number(&save, NULL, &obs->rad[i][id]);
    for (int id = 0; id < ctl->nd && ok; id++) ok = next_number(&save, NULL, &obs->tau[i][id]);
    if (!ok) continue;
    obs->nr++;
  }
  fclose(in);
  if (obs->nr < 1) DIE("Could not read any data!");
}

void write_obs(char const *filename, ctl_t const *ctl, obs_t const *obs) {
  if (ctl->checkmode) { printf("# skip writing target file name for observation data: %s\n", filename); return; }
  printf("Write observation data: %s\n", filename);
  FILE *out = fopen(filename, "w");
  if (!out) DIE("cannot write %s", filename);
  fprintf(out, "# $1 = time (seconds since 2000-01-01T00:00Z)\n"
               "# $2 = observer altitude [km]\n"
               "# $3 = observer longitude [deg]\n"
               "# $4 = observer latitude [deg]\n"
               "# $5 = view point altitude [km]\n"
               "# $6 = view point longitude [deg]\n"
               "# $7 = view point latitude [deg]\n"
               "# $8 = tangent point altitude [km]\n"
               "# $9 = tangent point longitude [deg]\n"
               "# $10 = tangent point latitude [deg]\n");
  int n = 10;
  char const *what = ctl->write_bbt ? "brightness temperature [K]" : "radiance [W/(m^2 sr cm^-1)]";
  for (int id = 0; id < ctl->nd; id++) fprintf(out, "# $%d = channel %g: %s\n", ++n, ctl->nu[id], what);
  for (int id = 0; id < ctl->nd; id++) {
    ++n;
    if ((ctl->nd < 65) || (id < 1) || (id > ctl->nd - 2)) fprintf(out, "# $%d = channel %g: transmittance\n", n, ctl->nu[id]);
    else if (1 == id) fprintf(out, "# $%d through $%d transmittance\n", n, n + ctl->nd - 3);
  }
  for (int ir = 0; ir < obs->nr; ir++) {
    if (ir == 0 || obs->time[ir] != obs->time[ir - 1]) fprintf(out, "\n");
    fprintf(out, "%.2f %g %g %g %g %g %g %g %g %g", obs->time[ir], obs->obsz[ir], obs->obslon[ir], obs->obslat[ir],
            obs->vpz[ir], obs->vplon[ir], obs->vplat[ir], obs->tpz[ir], obs->tplon[ir], obs->tplat[ir]);
    for (int id = 0; id < ctl->nd; id++) fprintf(out, " %g", obs->rad[ir][id]);
    for (int id = 0; id < ctl->nd; id++) fprintf(out, " %g", obs->tau[ir][id]);
    fprintf(out, "\n");
  }
  fclose(out);
}

void write_atm(char const *filename, ctl_t const *ctl, atm_t const *atm) {
  if (ctl->checkmode) { printf("# skip writing target file name for atmospheric data: %s\n", filename); return; }
  printf("Write atmospheric data: %s\n", filename);
  FILE *out = fopen(filename, "w");
  if (!out) DIE("cannot write %s", filename);
  fprintf(out, "# $1 = time (seconds since 2000-01-01T00:00Z)\n"
               "# $2 = altitude [km]\n"
               "# $3 = longitude [deg]\n"
               "# $4 = latitude [deg]\n"
               "# $5 = pressure [hPa]\n"
               "# $6 = temperature [K]\n");
  int n = 6;
  for (int ig = 0; ig < ctl->ng; ig++) fprintf(out, "# $%d = %s volume mixing ratio\n", ++n, ctl->emitter[ig]);
  for (int iw = 0; iw < ctl->nw; iw++) fprintf(out, "# $%d = window %d: extinction [1/km]\n", ++n, iw);
  for (int ip = 0; ip < atm->np; ip++) {
    if (ip == 0 || atm->time[ip] != atm->time[ip - 1]) fprintf(out, "\n");
    fprintf(out, "%.2f %g %g %g %g %g", atm->time[ip], atm->z[ip], atm->lon[ip], atm->lat[ip], atm->p[ip], atm->t[ip]);
    for (int ig = 0; ig < ctl->ng; ig++) fprintf(out, " %g", atm->q[ig][ip]);
    for (int iw = 0; iw < ctl->nw; iw++) fprintf(out, " %g", atm->k[iw][ip]);
    fprintf(out, "\n");
  }
  fclose(out);
}
