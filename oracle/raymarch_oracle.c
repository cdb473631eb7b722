/* placeholder so the Makefile links; replaced by the real float restatement */
int vro_placeholder(void) { return 0; }
