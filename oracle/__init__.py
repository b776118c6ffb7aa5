"""CPU oracle for the ConvLSTM / ST-LSTM hot path. TEST INFRASTRUCTURE ONLY — see oracle/vpx_oracle.c."""
