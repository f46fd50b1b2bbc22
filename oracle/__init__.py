"""TEST INFRASTRUCTURE ONLY.  See oracle/tardis_mc_oracle.c.  Never imported by the product (tardis_amd/)."""
