"""Fake front end that replays captured bytes through the reference's buffer hand-off.

It mirrors the cadence of USBH_RTLSDR_Process (Middlewares/ST/STM32_USB_Host_Library/Class/RTLSDR/Src/usbh_rtlsdr.c:
1058-1101): XFER_START submits one bulk-IN URB into the single reused buffer, XFER_WAIT polls until the URB is done,
XFER_COMPLETE is where the consumer runs before the FSM re-arms.  URB sizes honour the reference's limits: a multiple
of the 512-byte bulk max-packet (usbh_rtlsdr.c:206-207,230) and at most 65 535 bytes because USBH_BulkReceiveData takes
a uint16_t length (Core/Src/usbh_ioreq.c:218-233).
"""
import enum

import numpy as np


class XferState(enum.IntEnum):  # RTLSDR_xferStateTypeDef, usbh_rtlsdr.h:156-162
    START = 0
    WAIT = 1
    COMPLETE = 2


class ReplayFrontEnd:
    MAX_URB = 65535
    PACKET = 512

    def __init__(self, data: np.ndarray, buff_size=512, wait_polls=1):
        if buff_size % self.PACKET or not (0 < buff_size <= self.MAX_URB):
            raise ValueError("buffSize must be a multiple of 512 and <= 65535 (uint16_t URB length)")
        self.data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        self.buff = np.zeros(buff_size, dtype=np.uint8)   # CommItf.buff: ONE buffer, reused for every URB
        self.buff_size = buff_size                        # CommItf.buffSize
        self.state = XferState.START
        self.pos = 0
        self.last_xfer_size = 0                           # USBH_LL_GetLastXferSize
        self._wait_polls, self._polls = wait_polls, 0

    @property
    def exhausted(self):
        return self.pos >= self.data.size and self.state == XferState.START

    def process(self):
        """One call of the background process; returns the state AFTER the call (like polling xferState in main.c:76)."""
        if self.state == XferState.START:
            if self.pos >= self.data.size:
                return self.state
            self._polls = 0
            self.state = XferState.WAIT
        elif self.state == XferState.WAIT:
            self._polls += 1
            if self._polls >= self._wait_polls:
                n = min(self.buff_size, self.data.size - self.pos)
                self.buff[:n] = self.data[self.pos: self.pos + n]   # the IRQ-side FIFO copy (USB_ReadPacket)
                self.pos += n
                self.last_xfer_size = n
                self.state = XferState.COMPLETE
        else:  # COMPLETE: the reference re-arms immediately; the consumer must be done with buff by now
            self.state = XferState.START
        return self.state

    def run(self, consumer):
        """Drive the FSM to the end of the data, calling consumer(buff, last_xfer_size) in XFER_COMPLETE."""
        outs = []
        while not self.exhausted:
            if self.process() == XferState.COMPLETE:
                outs.append(consumer(self.buff, self.last_xfer_size))
                self.buff[:] = 0xEE   # prove the consumer did not keep a reference to the reused buffer
        return outs
