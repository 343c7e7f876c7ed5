/* oracle_constants.h -- the oracle's OWN copy of the physical constants on the path.
 *
 * TEST INFRASTRUCTURE (see jurassic_oracle.h).  Deliberately not shared with the product: the kernels and the
 * host code take their constants from include/jurassic_abi.h (JUR_C1 ...), the oracle takes them from here, so
 * that a wrong value on either side shows up as a parity failure instead of cancelling.  tests/test_abi_cpu.py
 * additionally holds both headers against literals it carries itself.
 *
 * Sources (reference tree, file:line):
 *   C1, C2, P0, RE   src/jurassic.h:109-126 -- literal #defines.
 *   N_A, k_B, R      taken by the reference from GSL 2.5 (lib/build.sh:15), not vendored in the reference tree:
 *                      GSL_CONST_NUM_AVOGADRO      gsl/gsl_const_num.h   used at src/jr_common.h:330
 *                      GSL_CONST_MKSA_BOLTZMANN    gsl/gsl_const_mksa.h  used at src/jr_common.h:450
 *                      GSL_CONST_MKSA_MOLAR_GAS    gsl/gsl_const_mksa.h  used at src/jr_common.h:744,757
 *                    GSL 2.5 publishes the CODATA values of its day: N_A = 6.02214199e23 /mol (CODATA 1998),
 *                    k_B = 1.3806504e-23 J/K and R = 8.314472 J/(K mol) (CODATA 2006).  They cannot be re-read
 *                    offline (no GSL in this image).  Against CODATA 2018 this k_B differs by 1.0e-6 relative: it
 *                    enters through the column densities only, but at the level of the 1e-6 contract -- which
 *                    is why the value is pinned by a test instead of being taken from a newer table.
 */
#ifndef ORACLE_CONSTANTS_H
#define ORACLE_CONSTANTS_H

#define ORC_C1 1.19104259e-8      /* src/jurassic.h:111  first spectroscopic constant 2 h c^2 [W/(m^2 sr cm^-4)] */
#define ORC_C2 1.43877506         /* src/jurassic.h:114  second spectroscopic constant h c / k [K/cm^-1]         */
#define ORC_P0 1013.25            /* src/jurassic.h:120  standard pressure [hPa]                                */
#define ORC_RE 6367.421           /* src/jurassic.h:126  mean radius of Earth [km]                              */
#define ORC_AVOGADRO  6.02214199e23   /* GSL 2.5 GSL_CONST_NUM_AVOGADRO   [1/mol]      */
#define ORC_BOLTZMANN 1.3806504e-23   /* GSL 2.5 GSL_CONST_MKSA_BOLTZMANN [J/K]        */
#define ORC_MOLAR_GAS 8.314472        /* GSL 2.5 GSL_CONST_MKSA_MOLAR_GAS [J/(K mol)]  */

#endif
