/* TEST-ONLY: see Rinternals.h in this directory. */
