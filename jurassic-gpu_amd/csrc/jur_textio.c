/* jur_textio.c -- see jur_textio.h */
#define _GNU_SOURCE
#include <math.h>
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

void read_ctl(int argc, char *argv[], ctl_t *ctl) {
  printf("\nJuelich Rapid Spectral Simulation Code (JURASSIC), MI355X forward model\n(executable: %s)\n\n", argv[0]);
  ctl->ng = (int)scan_ctl(argc, argv, "NG", -1, "0", NULL);
  if (ctl->ng < 0 || ctl->ng > JUR_NG) DIE("Set 0 <= NG <= %d", JUR_NG);
  for (int ig = 0; ig < ctl->ng; ig++) scan_ctl(argc, argv, "EMITTER", ig, "", ctl->emitter[ig]);
  ctl->nd = (int)scan_ctl(argc, argv, "ND", -1, "0", NULL);
  if (ctl->nd < 0 || ctl->nd > JUR_ND) DIE("Set 0 <= ND <= %d", JUR_ND);
  for (int id = 0; id < ctl->nd; id++) ctl->nu[id] = scan_ctl(argc, argv, "NU", id, "", NULL);
  ctl->nw = (int)scan_ctl(argc, argv, "NW", -1, "1", NULL);
  if (ctl->nw < 0 || ctl->nw > JUR_NW) DIE("Set 0 <= NW <= %d", JUR_NW);
  for (int id = 0; id < ctl->nd; id++) ctl->window[id] = (int)scan_ctl(argc, argv, "WINDOW", id, "0", NULL);
  scan_ctl(argc, argv, "TBLBASE", -1, "-", ctl->tblbase);
  ctl->hydz = scan_ctl(argc, argv, "HYDZ", -1, "-999", NULL);
  ctl->ctm_co2 = (int)scan_ctl(argc, argv, "CTM_CO2", -1, "1", NULL);
  ctl->ctm_h2o = (int)scan_ctl(argc, argv, "CTM_H2O", -1, "1", NULL);
  ctl->ctm_n2 = (int)scan_ctl(argc, argv, "CTM_N2", -1, "1", NULL);
  ctl->ctm_o2 = (int)scan_ctl(argc, argv, "CTM_O2", -1, "1", NULL);
  {  /* continua whose spectral range holds no channel are switched off (jurassic.c:954-968) */
    int in_co2 = 0, in_h2o = 0, in_n2 = 0, in_o2 = 0;
    for (int id = 0; id < ctl->nd; id++) {
      double const nu = ctl->nu[id];
      in_co2 += (nu < 4000);
      in_h2o += (nu < 20000);
      in_n2 += (nu >= 2120 && nu <= 2605);
      in_o2 += (nu >= 1360 && nu <= 1805);
    }
    if (0 == in_co2 && ctl->ctm_co2) { ctl->ctm_co2 = 0; printf("No frequency in CO2 range, automatically set CTM_CO2 = 0\n"); }
    if (0 == in_h2o && ctl->ctm_h2o) { ctl->ctm_h2o = 0; printf("No frequency in H2O range, automatically set CTM_H20 = 0\n"); }
    if (0 == in_n2 && ctl->ctm_n2) { ctl->ctm_n2 = 0; printf("No frequency in N2 range, automatically set CTM_N2 = 0\n"); }
    if (0 == in_o2 && ctl->ctm_o2) { ctl->ctm_o2 = 0; printf("No frequency in O2 range, automatically set CTM_O2 = 0\n"); }
  }
  ctl->ip = (int)scan_ctl(argc, argv, "IP", -1, "1", NULL);
  ctl->cz = scan_ctl(argc, argv, "CZ", -1, "0", NULL);
  ctl->cx = scan_ctl(argc, argv, "CX", -1, "0", NULL);
  ctl->refrac = (int)scan_ctl(argc, argv, "REFRAC", -1, "1", NULL);
  ctl->rayds = scan_ctl(argc, argv, "RAYDS", -1, "10", NULL);
  ctl->raydz = scan_ctl(argc, argv, "RAYDZ", -1, "0.5", NULL);
  scan_ctl(argc, argv, "FOV", -1, "-", ctl->fov);
  ctl->retp_zmin = scan_ctl(argc, argv, "RETP_ZMIN", -1, "-999", NULL);
  ctl->retp_zmax = scan_ctl(argc, argv, "RETP_ZMAX", -1, "-999", NULL);
  ctl->rett_zmin = scan_ctl(argc, argv, "RETT_ZMIN", -1, "-999", NULL);
  ctl->rett_zmax = scan_ctl(argc, argv, "RETT_ZMAX", -1, "-999", NULL);
  for (int ig = 0; ig < ctl->ng; ig++) {
    ctl->retq_zmin[ig] = scan_ctl(argc, argv, "RETQ_ZMIN", ig, "-999", NULL);
    ctl->retq_zmax[ig] = scan_ctl(argc, argv, "RETQ_ZMAX", ig, "-999", NULL);
  }
  for (int iw = 0; iw < ctl->nw; iw++) {
    ctl->retk_zmin[iw] = scan_ctl(argc, argv, "RETK_ZMIN", iw, "-999", NULL);
    ctl->retk_zmax[iw] = scan_ctl(argc, argv, "RETK_ZMAX", iw, "-999", NULL);
  }
  ctl->write_bbt = (int)scan_ctl(argc, argv, "WRITE_BBT", -1, "0", NULL);
  ctl->write_matrix = (int)scan_ctl(argc, argv, "WRITE_MATRIX", -1, "0", NULL);
  ctl->formod = (int)scan_ctl(argc, argv, "FORMOD", -1, "2", NULL);
  scan_ctl(argc, argv, "RFMBIN", -1, "-", ctl->rfmbin);
  scan_ctl(argc, argv, "RFMHIT", -1, "-", ctl->rfmhit);
  for (int ig = 0; ig < ctl->ng; ig++) scan_ctl(argc, argv, "RFMXSC", ig, "-", ctl->rfmxsc[ig]);
  ctl->useGPU = (int)scan_ctl(argc, argv, "USEGPU", -1, "0", NULL);
  ctl->checkmode = (int)scan_ctl(argc, argv, "CHECKMODE", -1, "0", NULL);
  printf("CHECKMODE = %d (%s)\n", ctl->checkmode, (0 == ctl->checkmode) ? "run" : ((ctl->checkmode > 0) ? "skip" : "obs"));
  ctl->read_binary = (int)scan_ctl(argc, argv, "READ_BINARY", -1, "-1", NULL);
  ctl->write_binary = (int)scan_ctl(argc, argv, "WRITE_BINARY", -1, "1", NULL);
  ctl->gpu_nbytes_shared_memory = (int)scan_ctl(argc, argv, "GPU_SHARED_MEMORY", -1, "0", NULL);
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
